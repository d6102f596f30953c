// crc32.hip -- K2 bgzf_crc32 for gfx950 (wave64): CRC32 of every inflated BGZF member against its trailer.
//
// Replaces the check noodles-bgzf 0.49.0 `Reader::read_block` performs after libdeflate has inflated a block
// (un-vendored dependency of the reference; call sites bio-format-bam/src/storage.rs:161-169, 285-295).
#include "kernels.h"

namespace bioscan {

// ---- K2: CRC32 (IEEE 802.3, reflected) of each inflated member vs its BGZF trailer ----------------
// noodles-bgzf verifies every block's CRC32 after inflating it; this is the same check.
// One lane per QUARTER of a member (16 KiB; the four lanes of a member are neighbours in a wave): slice-by-16 tables live
// in LDS (16 KiB per workgroup, built by the workgroup itself), each lane streams its part with aligned 16-byte loads,
// and lane 0 of the four joins the parts: the CRC register after a part B that follows a part A is
// Z_|B|(register after A) ^ (register of B started from 0), Z_n = "n zero bytes", a linear map applied as at most 15 32x32
// bit matrices (Z_1, Z_2, Z_4, ... by repeated squaring, also built by the workgroup: 1.9 KiB).  A lane per whole member
// (r01..r03) left the kernel with 655 360 equal pieces of work for 524 288 lane slots -- 1.25 rounds, the second one a
// quarter full; with 2.6 M pieces the rounds even out.  Measured on config 2: 9.59 -> 9.25 ms only -- the kernel is not
// HBM-bound but bound by its LDS lookups (one per byte, 64 random addresses per instruction: SQ_LDS_IDX_ACTIVE is 70 % of
// the kernel's cycles, two thirds of them bank conflicts; profiles/r03/k1_lds_pipe_summary.txt).
constexpr int CRC_T = 256;
constexpr int CRC_PARTS = 4;                 // lanes per member
constexpr uint32_t CRC_PART_LOG2 = 14;       // bytes per part = 16 KiB (4 parts cover the 64 KiB a member can hold)
__device__ __forceinline__ uint32_t crc_matvec(const uint32_t* m, uint32_t s) {
  uint32_t r = 0;
#pragma unroll
  for (int j = 0; j < 32; j++) r ^= m[j] & (0u - ((s >> j) & 1u));
  return r;
}
// NL (the FASTQ scan): K2 reads every inflated byte anyway, so it also counts the newlines -- per 16 KiB tile of the text buffer
// (tile t = bytes [16384 t, 16384 (t + 1)) of the buffer; a lane's part lies in at most two tiles, so at most two atomic adds
// per lane), which is what the count pass of the newline index (fastq_kernels.hip: k_nl_count, one more sweep over the text)
// would have produced.  nl_bias: buffer position of a member's byte = its inflated offset + nl_bias.
__device__ __forceinline__ uint32_t crc_nl16(const uint4& v) {
  auto m8 = [](uint64_t w) -> uint32_t {   // bytes of w equal to '\n' (exact SWAR zero-byte test)
    const uint64_t x = w ^ 0x0A0A0A0A0A0A0A0Aull;
    const uint64_t t = (x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full;
    return (uint32_t)__popcll(~(t | x | 0x7F7F7F7F7F7F7F7Full));
  };
  return m8((uint64_t)v.x | (uint64_t)v.y << 32) + m8((uint64_t)v.z | (uint64_t)v.w << 32);
}
template <bool NL>
__global__ __launch_bounds__(CRC_T) void k_bgzf_crc32(const uint8_t* __restrict__ comp,
                                                       const uint64_t* __restrict__ blk_coff,
                                                       const uint64_t* __restrict__ blk_uoff,
                                                       const uint8_t* __restrict__ out_all, uint32_t n_blocks,
                                                       uint32_t* status, uint32_t* __restrict__ store,
                                                       uint32_t* __restrict__ nl_cnt, uint64_t nl_bias) {
  __shared__ uint32_t T[16][256];
  __shared__ uint32_t Z[CRC_PART_LOG2 + 1][32];   // Z[k] = "2^k zero bytes": column j = image of bit j
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
  if (threadIdx.x < 32) {
    const uint32_t s = 1u << threadIdx.x;
    Z[0][threadIdx.x] = T[0][s & 0xFF] ^ (s >> 8);
  }
  for (uint32_t k = 1; k <= CRC_PART_LOG2; k++) {
    __syncthreads();
    if (threadIdx.x < 32) Z[k][threadIdx.x] = crc_matvec(Z[k - 1], Z[k - 1][threadIdx.x]);
  }
  __syncthreads();
  const uint32_t unit = blockIdx.x * CRC_T + threadIdx.x;
  const uint32_t b = unit / CRC_PARTS, q = unit % CRC_PARTS;
  const bool live = b < n_blocks;   // (no early return: the parts of a member are joined with wave shuffles below)
  uint32_t n_all = 0, n = 0;
  uint64_t nl_pos = 0;
  const uint8_t* p = out_all;
  if (live) {
    const uint64_t u0 = blk_uoff[b];
    n_all = (uint32_t)(blk_uoff[b + 1] - u0);
    const uint32_t lo = q << CRC_PART_LOG2;
    if (lo < n_all) { n = n_all - lo; if (n > (1u << CRC_PART_LOG2)) n = 1u << CRC_PART_LOG2; }
    p = out_all + u0 + lo;
    if constexpr (NL) nl_pos = u0 + lo + nl_bias;
  }
  const uint32_t part_len = n;
  uint32_t c = q == 0 ? 0xFFFFFFFFu : 0u;
  // (NL) first tile of the part, bytes of the part that lie in it, newlines in it / behind it
  const uint64_t nl_tile = nl_pos >> 14;
  const uint32_t nl_in_first = (uint32_t)(((nl_tile + 1) << 14) - nl_pos);
  uint32_t nl_a = 0, nl_b = 0, done = 0;
  while (n && ((uintptr_t)p & 15)) {
    if constexpr (NL) { if (*p == '\n') { if (done < nl_in_first) nl_a++; else nl_b++; } done++; }
    c = (c >> 8) ^ T[0][(c ^ *p++) & 0xFF]; n--;
  }
  // slice-by-16: one 16-byte load per step, 16 table lookups of which only 4 depend on the running CRC
  const uint4* w = (const uint4*)p;
  const uint32_t nq = n >> 4;
#define CRC_STEP16(v) do { \
    const uint32_t a0 = c ^ (v).x, a1 = (v).y, a2 = (v).z, a3 = (v).w; \
    c = T[15][a0 & 0xFF] ^ T[14][(a0 >> 8) & 0xFF] ^ T[13][(a0 >> 16) & 0xFF] ^ T[12][a0 >> 24] ^ \
        T[11][a1 & 0xFF] ^ T[10][(a1 >> 8) & 0xFF] ^ T[9][(a1 >> 16) & 0xFF] ^ T[8][a1 >> 24] ^ \
        T[7][a2 & 0xFF] ^ T[6][(a2 >> 8) & 0xFF] ^ T[5][(a2 >> 16) & 0xFF] ^ T[4][a2 >> 24] ^ \
        T[3][a3 & 0xFF] ^ T[2][(a3 >> 8) & 0xFF] ^ T[1][(a3 >> 16) & 0xFF] ^ T[0][a3 >> 24]; } while (0)
  // 128 bytes (one cache line of this lane's part) per outer step: the eight loads are issued together so the
  // line is consumed by one fill instead of being re-requested across iterations (the lanes in flight thrash L1).
  // (NL) the chunks in front of the part's tile boundary and those behind it run as two stretches, each with its own counter
  // (a 16-byte chunk never straddles a tile boundary: both are 16-byte aligned in the buffer -- the host checks that)
  auto stretch = [&](uint32_t k, const uint32_t k_end, uint32_t& nl_acc) {
    for (; k + 8 <= k_end; k += 8) {
      const uint4 v0 = w[k], v1 = w[k + 1], v2 = w[k + 2], v3 = w[k + 3], v4 = w[k + 4], v5 = w[k + 5], v6 = w[k + 6], v7 = w[k + 7];
      CRC_STEP16(v0); CRC_STEP16(v1); CRC_STEP16(v2); CRC_STEP16(v3);
      CRC_STEP16(v4); CRC_STEP16(v5); CRC_STEP16(v6); CRC_STEP16(v7);
      if constexpr (NL) nl_acc += crc_nl16(v0) + crc_nl16(v1) + crc_nl16(v2) + crc_nl16(v3) + crc_nl16(v4) + crc_nl16(v5) + crc_nl16(v6) + crc_nl16(v7);
    }
    for (; k < k_end; k++) {
      const uint4 v = w[k]; CRC_STEP16(v);
      if constexpr (NL) nl_acc += crc_nl16(v);
    }
  };
  if constexpr (NL) {
    const uint32_t k_split = done < nl_in_first ? ((nl_in_first - done) >> 4 < nq ? (nl_in_first - done) >> 4 : nq) : 0u;
    stretch(0, k_split, nl_a);
    stretch(k_split, nq, nl_b);
    done += nq * 16u;
  } else {
    uint32_t unused = 0;
    stretch(0, nq, unused);
  }
#undef CRC_STEP16
  p += (size_t)nq * 16;
  n &= 15;
  while (n--) {
    if constexpr (NL) { if (*p == '\n') { if (done < nl_in_first) nl_a++; else nl_b++; } done++; }
    c = (c >> 8) ^ T[0][(c ^ *p++) & 0xFF];
  }
  if constexpr (NL) {
    if (nl_a) atomicAdd(&nl_cnt[nl_tile], nl_a);
    if (nl_b) atomicAdd(&nl_cnt[nl_tile + 1], nl_b);
  }
  // join the parts in lane q == 0 of each group of four
  const int lane = threadIdx.x & 63;
  uint32_t s = c;
#pragma unroll
  for (int j = 1; j < CRC_PARTS; j++) {
    const uint32_t cj = (uint32_t)__shfl((int)c, (lane & ~(CRC_PARTS - 1)) + j, 64);
    const uint32_t lj = (uint32_t)__shfl((int)part_len, (lane & ~(CRC_PARTS - 1)) + j, 64);
    if (q == 0 && lj) {
      for (uint32_t k2 = 0; k2 <= CRC_PART_LOG2; k2++)
        if ((lj >> k2) & 1u) s = crc_matvec(Z[k2], s);
      s ^= cj;
    }
  }
  if (!live || q != 0) return;
  s ^= 0xFFFFFFFFu;
  if (store) store[b] = s;  // write path: the value that goes into the trailer of member b
  else {
    // validation mode: the CRC32 field of the member's trailer
    const uint8_t* tr = comp + blk_coff[b + 1] - 8;
    const uint32_t want = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
    if (s != want && status[b] == INF_OK) status[b] = INF_CRC_MISMATCH;
  }
}
void launch_bgzf_crc32(const uint8_t* comp, const uint64_t* blk_coff, const uint64_t* blk_uoff, const uint8_t* out,
                       uint32_t n_blocks, uint32_t* status, hipStream_t st, uint32_t* nl_cnt, uint64_t nl_bias) {
  if (!n_blocks) return;
  const dim3 g((uint32_t)(((uint64_t)n_blocks * CRC_PARTS + CRC_T - 1) / CRC_T));
  if (nl_cnt)
    hipLaunchKernelGGL(k_bgzf_crc32<true>, g, dim3(CRC_T), 0, st, comp, blk_coff, blk_uoff, out, n_blocks, status, (uint32_t*)nullptr, nl_cnt, nl_bias);
  else
    hipLaunchKernelGGL(k_bgzf_crc32<false>, g, dim3(CRC_T), 0, st, comp, blk_coff, blk_uoff, out, n_blocks, status, (uint32_t*)nullptr,
                       (uint32_t*)nullptr, (uint64_t)0);
}
// write path: crc[b] = CRC32 of payload[off[b] .. off[b + 1])
void launch_crc32_store(const uint8_t* payload, const uint64_t* off, uint32_t n_members, uint32_t* crc, hipStream_t st) {
  if (!n_members) return;
  hipLaunchKernelGGL(k_bgzf_crc32<false>, dim3((uint32_t)(((uint64_t)n_members * CRC_PARTS + CRC_T - 1) / CRC_T)), dim3(CRC_T), 0, st, (const uint8_t*)nullptr, (const uint64_t*)nullptr,
                     off, payload, n_members, (uint32_t*)nullptr, crc, (uint32_t*)nullptr, (uint64_t)0);
}

}  // namespace bioscan
