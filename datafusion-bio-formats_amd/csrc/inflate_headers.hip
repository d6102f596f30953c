// inflate_headers.hip -- K0: the dynamic-Huffman header of every BGZF member's FIRST DEFLATE block, one member per LANE.
//
// Inside K1 (inflate_v3.hip) a block header is parsed by the whole wave on behalf of one member: the code-length
// sequence of RFC 1951 3.2.7 is a serial loop of ~250 iterations (one per code-length symbol) whose every
// instruction serves one member -- 7.5 % of K1's wave cycles for config 2 (profiles/r03, BIOSCAN_DEBUG anatomy).  A
// BGZF member almost always holds ONE block, and its header starts at a known place (the first payload bit), so all first
// headers of a launch can be parsed up front with 64 members per wave: the same ~250 iterations then serve 64 members.
// K1 picks the result up (flags, bit position behind the header, the 320 code lengths as nibbles) and goes straight to
// its table build; every later block of a member, and any header this kernel does not like, takes K1's own parser --
// which is also the one that reports errors, so an invalid header is simply left to it (flag 0).
// Replaces (with K1) libdeflate's `read_dynamic_huffman_header` part of `deflate_decompress` (un-vendored dependency
// of the reference; call sites bio-format-bam/src/storage.rs:161-169).
#include "kernels.h"

namespace bioscan {

namespace {
constexpr int H_T = 256;                 // threads per workgroup = members per workgroup
// per-lane bit reader over the member's payload (unaligned 4-byte loads; the compressed image is padded behind its end)
struct LBits {
  const uint8_t* p;
  uint64_t bb;
  int bc;
  uint32_t taken;   // bits consumed since the payload's first bit
};
__device__ __forceinline__ void lb_refill(LBits& s) {
  if (s.bc <= 32) {
    uint32_t w;
    __builtin_memcpy(&w, s.p, 4);
    s.p += 4;
    s.bb |= (uint64_t)w << s.bc;
    s.bc += 32;
  }
}
__device__ __forceinline__ uint32_t lb_take(LBits& s, int n) {
  const uint32_t v = (uint32_t)s.bb & ((1u << n) - 1u);
  s.bb >>= n;
  s.bc -= n;
  s.taken += (uint32_t)n;
  return v;
}
__device__ __forceinline__ uint32_t rev_bits(uint32_t v, int n) { return __brev(v) >> (32 - n); }
}  // namespace

// rec: V3_PRE_DWORDS dwords per member: [0] bit 0 = usable, bit 1 = BFINAL; [1] bits of the payload consumed by the header
// (block header + code lengths); [2 .. 41] the 320 code lengths of K1's V3Build::lens (literal/length j at j, distance j at
// 288 + j, zero where the header names none), 4 bits each, 8 per dword, low nibble first.
__global__ __launch_bounds__(H_T) void k_bgzf_headers(const uint8_t* __restrict__ comp, const uint64_t* __restrict__ blk_coff, uint32_t n_blocks,
                                                       uint32_t* __restrict__ rec) {
  // the 128-entry decode table of the code-length code, one column per lane (entry = symbol << 3 | length, 0 = no code)
  __shared__ uint8_t T[128][H_T];
  const uint32_t b = blockIdx.x * H_T + threadIdx.x;
  if (b >= n_blocks) return;
  uint32_t* r = rec + (size_t)b * V3_PRE_DWORDS;
  r[0] = 0;   // not usable until proven otherwise
  const uint64_t coff = blk_coff[b], cend = blk_coff[b + 1];
  const uint8_t* hdr = comp + coff;
  const uint32_t magic = (uint32_t)hdr[0] | ((uint32_t)hdr[1] << 8) | ((uint32_t)hdr[2] << 16) | ((uint32_t)hdr[3] << 24);
  if ((magic & 0x04FFFFFFu) != 0x04088B1Fu) return;
  const uint32_t xlen = (uint32_t)hdr[10] | ((uint32_t)hdr[11] << 8);
  if (cend - coff < 12ull + xlen + 8ull) return;
  const uint64_t payload_bits = ((cend - coff) - 12 - xlen - 8) * 8;
  LBits in{hdr + 12 + xlen, 0, 0, 0};
  lb_refill(in);
  const uint32_t bfinal = lb_take(in, 1);
  if (lb_take(in, 2) != 2) return;   // stored / fixed / reserved: K1's own code
  const uint32_t hlit = lb_take(in, 5) + 257, hdist = lb_take(in, 5) + 1, hclen = lb_take(in, 4) + 4;
  if (hlit > 286 || hdist > 30) return;
  // code-length code: 19 lengths of 3 bits, packed into one register (symbol s at bits 3 s)
  uint64_t pl = 0;
  {
    const uint64_t ord0 = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
    const uint64_t ord1 = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
    for (uint32_t i = 0; i < hclen; i++) {
      lb_refill(in);
      const uint64_t v = lb_take(in, 3);
      const uint32_t s = (uint32_t)((i < 12 ? ord0 >> (5 * i) : ord1 >> (5 * (i - 12))) & 31u);
      pl |= v << (3 * s);
    }
  }
  // canonical codes of the code-length code; the code space must be exactly full, or one 1-bit code (libdeflate's rule)
  uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = 0; s < 19; s++) cnt[(pl >> (3 * s)) & 7]++;
  uint32_t first[8], code = 0, used = 0, nsym = 0;
  first[0] = 0;
  for (int l = 1; l <= 7; l++) {
    code <<= 1;
    first[l] = code;
    code += cnt[l];
    used += cnt[l] << (7 - l);
    nsym += cnt[l];
  }
  if (used > 128u || (used < 128u && !(nsym == 1 && cnt[1] == 1))) return;
  for (int i = 0; i < 128; i++) T[i][threadIdx.x] = 0;
  for (int s = 0; s < 19; s++) {
    const uint32_t l = (uint32_t)(pl >> (3 * s)) & 7u;
    if (!l) continue;
    const uint32_t rc = rev_bits(first[l]++, (int)l);
    for (uint32_t i = rc; i < 128u; i += 1u << l) T[i][threadIdx.x] = (uint8_t)((uint32_t)s << 3 | l);
  }
  // the hlit + hdist code lengths, streamed out as nibbles in slot order (literal/length j -> j, distance j -> 288 + j)
  const uint32_t total = hlit + hdist;
  uint32_t i = 0, prev = 0, cur_dw = 0, acc = 0;
  auto emit = [&](uint32_t seq, uint32_t val) {
    const uint32_t slot = seq < hlit ? seq : 288u + (seq - hlit);
    const uint32_t dw = slot >> 3;
    while (cur_dw < dw) { r[2 + cur_dw] = acc; acc = 0; cur_dw++; }
    acc |= val << ((slot & 7u) * 4u);
  };
  while (i < total) {
    lb_refill(in);
    const uint32_t e = T[(uint32_t)in.bb & 127u][threadIdx.x];
    const uint32_t l = e & 7u, sym = e >> 3;
    if (l == 0) return;
    lb_take(in, (int)l);
    if (sym < 16) {
      emit(i, sym);
      prev = sym;
      i++;
    } else {
      uint32_t rep, val;
      if (sym == 16) { if (i == 0) return; rep = 3 + lb_take(in, 2); val = prev; }
      else if (sym == 17) { rep = 3 + lb_take(in, 3); val = 0; }
      else { rep = 11 + lb_take(in, 7); val = 0; }
      if (i + rep > total) return;
      if (val) for (uint32_t k = 0; k < rep; k++) emit(i + k, val);
      if (sym != 16) prev = 0;
      i += rep;
    }
  }
  while (cur_dw < 40u) { r[2 + cur_dw] = acc; acc = 0; cur_dw++; }
  if ((uint64_t)in.taken > payload_bits) return;   // ran past the member: not a header K1 should trust
  r[1] = in.taken;
  r[0] = 1u | (bfinal << 1);
}

void launch_bgzf_headers(const uint8_t* comp, const uint64_t* blk_coff, uint32_t n_blocks, uint32_t* rec, hipStream_t st) {
  if (!n_blocks) return;
  hipLaunchKernelGGL(k_bgzf_headers, dim3((n_blocks + H_T - 1) / H_T), dim3(H_T), 0, st, comp, blk_coff, n_blocks, rec);
}

}  // namespace bioscan
