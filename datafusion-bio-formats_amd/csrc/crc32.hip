// crc32.hip -- K2 bgzf_crc32 for gfx950 (wave64): CRC32 of every inflated BGZF member against its trailer.
//
// Replaces the check noodles-bgzf 0.49.0 `Reader::read_block` performs after libdeflate has inflated a block
// (un-vendored dependency of the reference; call sites bio-format-bam/src/storage.rs:161-169, 285-295).
#include "kernels.h"

namespace bioscan {

// ---- K2: CRC32 (IEEE 802.3, reflected) of each inflated member vs its BGZF trailer ----------------
// noodles-bgzf verifies every block's CRC32 after inflating it; this is the same check.
// One lane per member (64 members per wave): slice-by-16 tables live in LDS (16 KiB per
// workgroup, built by the workgroup itself), each lane streams its member with aligned dword
// loads.  No cross-lane combine is needed, so the kernel is a plain table-driven CRC whose
// throughput comes from having ~650 k members in flight.
constexpr int CRC_T = 256;
__global__ __launch_bounds__(CRC_T) void k_bgzf_crc32(const uint8_t* __restrict__ comp,
                                                       const uint64_t* __restrict__ blk_coff,
                                                       const uint64_t* __restrict__ blk_uoff,
                                                       const uint8_t* __restrict__ out_all, uint32_t n_blocks,
                                                       uint32_t* status, uint32_t* __restrict__ store) {
  __shared__ uint32_t T[16][256];
  {
    uint32_t c = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    T[0][threadIdx.x] = c;
  }
  __syncthreads();
  {
    uint32_t c = T[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 16; k++) { c = (c >> 8) ^ T[0][c & 0xFF]; T[k][threadIdx.x] = c; }
  }
  __syncthreads();
  const uint32_t b = blockIdx.x * CRC_T + threadIdx.x;
  if (b >= n_blocks) return;
  const uint8_t* p = out_all + blk_uoff[b];
  uint32_t n = (uint32_t)(blk_uoff[b + 1] - blk_uoff[b]);
  uint32_t want = 0;
  if (!store) {  // validation mode: the CRC32 field of the member's trailer
    const uint8_t* tr = comp + blk_coff[b + 1] - 8;
    want = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
  }
  uint32_t c = 0xFFFFFFFFu;
  while (n && ((uintptr_t)p & 15)) { c = (c >> 8) ^ T[0][(c ^ *p++) & 0xFF]; n--; }
  // slice-by-16: one 16-byte load per step, 16 table lookups of which only 4 depend on the running CRC
  const uint4* w = (const uint4*)p;
  const uint32_t nq = n >> 4;
#define CRC_STEP16(v) do { \
    const uint32_t a0 = c ^ (v).x, a1 = (v).y, a2 = (v).z, a3 = (v).w; \
    c = T[15][a0 & 0xFF] ^ T[14][(a0 >> 8) & 0xFF] ^ T[13][(a0 >> 16) & 0xFF] ^ T[12][a0 >> 24] ^ \
        T[11][a1 & 0xFF] ^ T[10][(a1 >> 8) & 0xFF] ^ T[9][(a1 >> 16) & 0xFF] ^ T[8][a1 >> 24] ^ \
        T[7][a2 & 0xFF] ^ T[6][(a2 >> 8) & 0xFF] ^ T[5][(a2 >> 16) & 0xFF] ^ T[4][a2 >> 24] ^ \
        T[3][a3 & 0xFF] ^ T[2][(a3 >> 8) & 0xFF] ^ T[1][(a3 >> 16) & 0xFF] ^ T[0][a3 >> 24]; } while (0)
  uint32_t k = 0;
  // 128 bytes (one cache line of this lane's member) per outer step: the eight loads are issued together so the
  // line is consumed by one fill instead of being re-requested across iterations (the 655 k lanes in flight thrash L1)
  for (; k + 8 <= nq; k += 8) {
    const uint4 v0 = w[k], v1 = w[k + 1], v2 = w[k + 2], v3 = w[k + 3], v4 = w[k + 4], v5 = w[k + 5], v6 = w[k + 6], v7 = w[k + 7];
    CRC_STEP16(v0); CRC_STEP16(v1); CRC_STEP16(v2); CRC_STEP16(v3);
    CRC_STEP16(v4); CRC_STEP16(v5); CRC_STEP16(v6); CRC_STEP16(v7);
  }
  for (; k < nq; k++) { const uint4 v = w[k]; CRC_STEP16(v); }
#undef CRC_STEP16
  p += (size_t)nq * 16;
  n &= 15;
  while (n--) c = (c >> 8) ^ T[0][(c ^ *p++) & 0xFF];
  c ^= 0xFFFFFFFFu;
  if (store) store[b] = c;  // write path: the value that goes into the trailer of member b
  else if (c != want && status[b] == INF_OK) status[b] = INF_CRC_MISMATCH;
}
void launch_bgzf_crc32(const uint8_t* comp, const uint64_t* blk_coff, const uint64_t* blk_uoff, const uint8_t* out,
                       uint32_t n_blocks, uint32_t* status, hipStream_t st) {
  if (!n_blocks) return;
  hipLaunchKernelGGL(k_bgzf_crc32, dim3((n_blocks + CRC_T - 1) / CRC_T), dim3(CRC_T), 0, st, comp, blk_coff, blk_uoff, out, n_blocks, status,
                     (uint32_t*)nullptr);
}
// write path: crc[b] = CRC32 of payload[off[b] .. off[b + 1])
void launch_crc32_store(const uint8_t* payload, const uint64_t* off, uint32_t n_members, uint32_t* crc, hipStream_t st) {
  if (!n_members) return;
  hipLaunchKernelGGL(k_bgzf_crc32, dim3((n_members + CRC_T - 1) / CRC_T), dim3(CRC_T), 0, st, (const uint8_t*)nullptr, (const uint64_t*)nullptr,
                     off, payload, n_members, (uint32_t*)nullptr, crc);
}

}  // namespace bioscan
