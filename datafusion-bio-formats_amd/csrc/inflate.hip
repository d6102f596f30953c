// inflate.hip -- K1 bgzf_inflate and K2 bgzf_crc32 for gfx950 (wave64).
//
// Replaces noodles-bgzf 0.49.0 `Reader::read_block` + libdeflate `deflate_decompress`
// (un-vendored dependency of the reference; call sites bio-format-bam/src/storage.rs:161-169,
// 285-295).  Format: SAM spec 4.1 (BGZF member) + RFC 1951 (DEFLATE).
//
// Work decomposition: one BGZF member per 64-lane wavefront (workgroup = one wave, so
// every barrier is wave-local).  Huffman decode tables live in LDS (3.7 KB per wave); the
// compressed stream is staged 256 B at a time in a VGPR (lane i holds dword i) and read
// with v_readlane, so the serial bit reader never waits on LDS or HBM.  Literals are
// stored straight to the member's output range; LZ77 matches are queued one per lane
// (64 per batch) and resolved by the whole wave at once: a batch is independent of
// itself unless a match's source range overlaps the destination of an earlier match in
// the same batch, which is detected exactly and resolved in dependency order.
#include "kernels.h"

namespace bioscan {

#define WAVE 64
constexpr int LIT_BITS = 10;
constexpr int DIST_BITS = 8;

struct __attribute__((aligned(16))) InflateLds {
  uint16_t lit_fast[1 << LIT_BITS];   // sym << 4 | len ; 0 = code longer than LIT_BITS
  uint16_t dist_fast[1 << DIST_BITS];
  uint16_t lit_sorted[288];           // symbols ordered by (len, sym)
  uint16_t dist_sorted[32];
  uint16_t lit_count[16];
  uint16_t dist_count[16];
  uint8_t lens[320];                  // code lengths litlen (0..287) then dist (288..319)
  uint8_t pre_fast[128];              // sym << 3 | len (code-length code, <= 7 bits)
  uint8_t pre_lens[19];
  uint16_t t_offs[16];                // table-build scratch: first sorted index per length
  uint16_t t_first[16];               // first canonical code per length
  uint16_t t_w[16];
};

struct BitIn {
  const uint32_t* base;  // 4-byte aligned pointer at/before payload start
  uint32_t cur, nxt;     // staged dwords: lane i of `cur` holds base[cidx*64 + i]
  uint32_t cidx;
  uint32_t wpos;         // next dword index to consume
  uint64_t bb;
  int bc;
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ void bitin_init(BitIn& s, const uint8_t* p, int lane) {
  uintptr_t a = (uintptr_t)p;
  s.base = (const uint32_t*)(a & ~(uintptr_t)3);
  s.cidx = 0;
  s.cur = s.base[lane];
  s.nxt = s.base[64 + lane];
  s.wpos = 0;
  s.bb = 0;
  s.bc = 0;
  // consume first word and drop the misaligned low bytes
  uint32_t w = __builtin_amdgcn_readlane(s.cur, 0);
  s.wpos = 1;
  int skip = (int)(a & 3) * 8;
  s.bb = (uint64_t)(w >> skip);
  s.bc = 32 - skip;
}

__device__ __forceinline__ uint32_t bitin_next_word(BitIn& s, int lane) {
  uint32_t c = s.wpos >> 6;
  if (c != s.cidx) {
    s.cur = s.nxt;
    s.cidx = c;
    s.nxt = s.base[(size_t)(c + 1) * 64 + lane];
  }
  uint32_t w = __builtin_amdgcn_readlane(s.cur, s.wpos & 63);
  s.wpos++;
  return w;
}

// guarantee >= 33 valid bits
__device__ __forceinline__ void bitin_refill(BitIn& s, int lane) {
  if (s.bc <= 32) {
    s.bb |= (uint64_t)bitin_next_word(s, lane) << s.bc;
    s.bc += 32;
  }
}
__device__ __forceinline__ uint32_t bitin_take(BitIn& s, int n) {
  uint32_t v = (uint32_t)s.bb & ((1u << n) - 1u);
  s.bb >>= n;
  s.bc -= n;
  return v;
}
// byte position (relative to the aligned base) of the next unread bit, rounded down
__device__ __forceinline__ uint64_t bitin_bytepos(const BitIn& s) {
  return (uint64_t)s.wpos * 4 - (uint64_t)(s.bc >> 3);
}

__device__ __forceinline__ uint32_t bitrev(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// Build fast + canonical tables for one alphabet.  lens[0..n) in LDS.
// fast[] has 1<<fast_bits entries; sorted[] receives symbols ordered by (len, sym); count[1..15].
// Returns 0 on success, nonzero if the code is over-subscribed.
__device__ int build_tables(const uint8_t* lens, int n, uint16_t* fast, int fast_bits, uint16_t* sorted,
                            uint16_t* count, uint16_t* t_offs, uint16_t* t_first, uint16_t* t_w, int lane) {
  __syncthreads();
  for (int i = lane; i < (1 << fast_bits); i += WAVE) fast[i] = 0;
  if (lane < 16) count[lane] = 0;
  __syncthreads();
  if (lane == 0) {
    for (int s = 0; s < n; s++) count[lens[s]]++;
    count[0] = 0;
    uint32_t o = 0, code = 0;
    int left = 1, over = 0;
    for (int l = 1; l <= 15; l++) {
      uint32_t c = count[l];
      code <<= 1;
      t_first[l] = (uint16_t)code;
      t_offs[l] = (uint16_t)o;
      t_w[l] = (uint16_t)o;
      o += c;
      code += c;
      left <<= 1;
      left -= (int)c;
      if (left < 0) over = 1;
    }
    t_offs[0] = (uint16_t)o;   // total coded symbols
    t_first[0] = (uint16_t)over;
    for (int s = 0; s < n; s++) {
      int l = lens[s];
      if (l) sorted[t_w[l]++] = (uint16_t)s;
    }
  }
  __syncthreads();
  if (uni(t_first[0])) return 1;
  const uint32_t o = uni(t_offs[0]);
  // fill the fast table: lane-parallel over sorted symbols
  for (uint32_t k = lane; k < o; k += WAVE) {
    int sym = sorted[k];
    int l = lens[sym];
    if (l <= fast_bits) {
      uint32_t c = (uint32_t)t_first[l] + (k - t_offs[l]);
      uint32_t r = bitrev(c, l);
      uint16_t e = (uint16_t)((sym << 4) | l);
      for (uint32_t i = r; i < (1u << fast_bits); i += (1u << l)) fast[i] = e;
    }
  }
  __syncthreads();
  return 0;
}

// canonical slow path: decode one symbol bit by bit (codes longer than the fast table)
__device__ __forceinline__ int slow_decode(BitIn& s, const uint16_t* sorted, const uint16_t* count) {
  int code = 0, first = 0, index = 0;
  uint64_t bb = s.bb;
  for (int l = 1; l <= 15; l++) {
    code |= (int)(bb & 1);
    bb >>= 1;
    const int c = (int)uni(count[l]);
    if (code - c < first) {
      s.bb = bb;
      s.bc -= l;
      return (int)uni(sorted[index + (code - first)]);
    }
    index += c;
    first += c;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

// Resolve a batch of <= 64 queued matches (one per lane) against the output window in HBM.
__device__ void resolve_batch(uint8_t* out, int lane, int nm, uint32_t m_dst, uint32_t m_len, uint32_t m_dist) {
  const bool valid = lane < nm;
  const uint32_t src_lo = m_dst - m_dist;
  const uint32_t src_end = src_lo + m_len;
  const uint32_t src_hi = src_end < m_dst ? src_end : m_dst;  // part of the source that precedes own dst
  const uint32_t first_dst = __builtin_amdgcn_readlane(m_dst, 0);
  uint64_t dep = 0;
  const bool maybe = valid && src_hi > first_dst;
  if (__ballot(maybe) != 0ull) {
    for (int i = 0; i < nm - 1; i++) {
      uint32_t di = __builtin_amdgcn_readlane(m_dst, i);
      uint32_t li = __builtin_amdgcn_readlane(m_len, i);
      if (maybe && i < lane && di < src_hi && di + li > src_lo) dep |= 1ull << i;
    }
  }
  const uint64_t all = nm >= 64 ? ~0ull : ((1ull << nm) - 1ull);
  uint64_t done = 0;
  while (done != all) {
    const bool ready = valid && !((done >> lane) & 1ull) && ((dep & ~done) == 0ull);
    if (ready) {
      uint8_t* d = out + m_dst;
      const uint8_t* s = out + src_lo;
      if (m_dist >= 4) {
        uint32_t k = 0;
        for (; k + 4 <= m_len; k += 4) {
          uint8_t b0 = s[k], b1 = s[k + 1], b2 = s[k + 2], b3 = s[k + 3];
          d[k] = b0; d[k + 1] = b1; d[k + 2] = b2; d[k + 3] = b3;
        }
        for (; k < m_len; k++) d[k] = s[k];
      } else {
        for (uint32_t k = 0; k < m_len; k++) d[k] = s[k];
      }
    }
    // make this round's stores visible to the next round's loads (same wave, in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    done |= __ballot(ready);
  }
}

__global__ __launch_bounds__(WAVE) void k_bgzf_inflate(const uint8_t* __restrict__ comp,
                                                        const uint64_t* __restrict__ blk_coff,
                                                        const uint64_t* __restrict__ blk_uoff,
                                                        uint8_t* out_all, uint32_t n_blocks,
                                                        uint32_t* __restrict__ status) {
  __shared__ InflateLds L;
  const int lane = threadIdx.x;
  const uint32_t b = blockIdx.x;
  if (b >= n_blocks) return;
  const uint64_t coff = blk_coff[b];
  const uint64_t cend = blk_coff[b + 1];
  const uint8_t* hdr = comp + coff;
  uint8_t* out = out_all + blk_uoff[b];
  const uint32_t isize = (uint32_t)(blk_uoff[b + 1] - blk_uoff[b]);
  uint32_t st = INF_OK;

  // gzip member header: 1f 8b 08 04 .... XLEN ; payload starts at 12 + XLEN
  const uint32_t xlen = uni((uint32_t)hdr[10] | ((uint32_t)hdr[11] << 8));
  const uint32_t magic = uni((uint32_t)hdr[0] | ((uint32_t)hdr[1] << 8) | ((uint32_t)hdr[2] << 16) | ((uint32_t)hdr[3] << 24));
  if ((magic & 0x04FFFFFFu) != 0x04088B1Fu) {
    if (lane == 0) status[b] = INF_BAD_HEADER;
    return;
  }
  const uint8_t* payload = hdr + 12 + xlen;
  uint64_t payload_len = (cend - coff) - 12 - xlen - 8;

  BitIn in;
  bitin_init(in, payload, lane);
  uint64_t base_skew = (uint64_t)((uintptr_t)payload & 3);

  uint32_t opos = 0;
  uint32_t m_dst = 0, m_len = 0, m_dist = 0;  // queued matches (lane k = k-th of the batch)
  int nm = 0;
  int fixed_built = 0;

  for (;;) {
    bitin_refill(in, lane);
    const uint32_t bfinal = bitin_take(in, 1);
    const uint32_t btype = bitin_take(in, 2);
    if (btype == 0) {
      // stored: flush queued matches first (they may be sources), then raw copy
      if (nm) { resolve_batch(out, lane, nm, m_dst, m_len, m_dist); nm = 0; }
      bitin_take(in, in.bc & 7);  // to byte boundary
      uint64_t bytepos = bitin_bytepos(in) - base_skew;  // relative to payload
      const uint8_t* p = payload + bytepos;
      uint32_t len = uni((uint32_t)p[0] | ((uint32_t)p[1] << 8));
      uint32_t nlen = uni((uint32_t)p[2] | ((uint32_t)p[3] << 8));
      if ((len ^ 0xFFFFu) != nlen) { st = INF_BAD_STORED; break; }
      if (opos + len > isize || bytepos + 4 + len > payload_len) { st = INF_OVERRUN; break; }
      p += 4;
      for (uint32_t k = lane; k < len; k += WAVE) out[opos + k] = p[k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      opos += len;
      // re-anchor the bit reader (and the payload-relative bookkeeping) right after the raw bytes
      payload_len -= bytepos + 4 + len;
      payload = p + len;
      base_skew = (uint64_t)((uintptr_t)payload & 3);
      bitin_init(in, payload, lane);
      if (bfinal) break;
      continue;
    }
    if (btype == 3) { st = INF_BAD_BTYPE; break; }

    if (btype == 1) {
      // fixed Huffman code (RFC 1951 3.2.6)
      for (int i = lane; i < 320; i += WAVE) {
        uint8_t l;
        if (i < 144) l = 8; else if (i < 256) l = 9; else if (i < 280) l = 7; else if (i < 288) l = 8; else l = 5;
        L.lens[i] = l;
      }
      __syncthreads();
      build_tables(L.lens, 288, L.lit_fast, LIT_BITS, L.lit_sorted, L.lit_count, L.t_offs, L.t_first, L.t_w, lane);
      build_tables(L.lens + 288, 32, L.dist_fast, DIST_BITS, L.dist_sorted, L.dist_count, L.t_offs, L.t_first, L.t_w, lane);
      (void)fixed_built;
    } else {
      // dynamic Huffman code (RFC 1951 3.2.7)
      bitin_refill(in, lane);
      const uint32_t hlit = bitin_take(in, 5) + 257;
      const uint32_t hdist = bitin_take(in, 5) + 1;
      const uint32_t hclen = bitin_take(in, 4) + 4;
      if (hlit > 286 || hdist > 30) { st = INF_BAD_CODE | (1u << 8); break; }
      if (lane < 19) L.pre_lens[lane] = 0;
      __syncthreads();
      {
        const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (uint32_t i = 0; i < hclen; i++) {
          bitin_refill(in, lane);
          uint32_t v = bitin_take(in, 3);
          if (lane == 0) L.pre_lens[order[i]] = (uint8_t)v;
        }
      }
      __syncthreads();
      // precode fast table (7 bits)
      {
        for (int i = lane; i < 128; i += WAVE) L.pre_fast[i] = 0;
        __syncthreads();
        if (lane == 0) {
          uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (int s = 0; s < 19; s++) cnt[L.pre_lens[s]]++;
          cnt[0] = 0;
          uint32_t next[8];
          uint32_t code = 0;
          for (int l = 1; l <= 7; l++) { code = (code + cnt[l - 1]) << 1; next[l] = code; }
          for (int s = 0; s < 19; s++) {
            int l = L.pre_lens[s];
            if (!l) continue;
            uint32_t r = bitrev(next[l]++, l);
            for (uint32_t i = r; i < 128; i += (1u << l)) L.pre_fast[i] = (uint8_t)((s << 3) | l);
          }
        }
        __syncthreads();
      }
      // code lengths
      {
        const uint32_t total = hlit + hdist;
        uint32_t i = 0;
        uint32_t prev = 0;
        int bad = 0;
        while (i < total) {
          bitin_refill(in, lane);
          uint32_t e = uni(L.pre_fast[(uint32_t)in.bb & 127]);
          uint32_t l = e & 7, sym = e >> 3;
          if (l == 0) { bad = 1; break; }
          bitin_take(in, l);
          if (sym < 16) {
            if (lane == 0) L.lens[i < hlit ? i : 288 + (i - hlit)] = (uint8_t)sym;
            prev = sym;
            i++;
          } else {
            uint32_t rep, val;
            if (sym == 16) { if (i == 0) { bad = 1; break; } rep = 3 + bitin_take(in, 2); val = prev; }
            else if (sym == 17) { rep = 3 + bitin_take(in, 3); val = 0; }
            else { rep = 11 + bitin_take(in, 7); val = 0; }
            if (i + rep > total) { bad = 1; break; }
            if (lane == 0)
              for (uint32_t k = 0; k < rep; k++) {
                uint32_t j = i + k;
                L.lens[j < hlit ? j : 288 + (j - hlit)] = (uint8_t)val;
              }
            if (sym != 16) prev = 0;
            i += rep;
          }
        }
        if (bad) { st = INF_BAD_CODE | (2u << 8); break; }
        // zero the unused tails
        for (uint32_t k = hlit + lane; k < 288; k += WAVE) L.lens[k] = 0;
        for (uint32_t k = 288 + hdist + lane; k < 320; k += WAVE) L.lens[k] = 0;
        __syncthreads();
      }
      if (build_tables(L.lens, 288, L.lit_fast, LIT_BITS, L.lit_sorted, L.lit_count, L.t_offs, L.t_first, L.t_w, lane)) { st = INF_BAD_CODE | (3u << 8); break; }
      if (build_tables(L.lens + 288, 32, L.dist_fast, DIST_BITS, L.dist_sorted, L.dist_count, L.t_offs, L.t_first, L.t_w, lane)) { st = INF_BAD_CODE | (4u << 8); break; }
    }

    // ---- symbol loop ----
    int err = 0;
    for (;;) {
      bitin_refill(in, lane);
      uint32_t e = uni(L.lit_fast[(uint32_t)in.bb & ((1u << LIT_BITS) - 1u)]);
      int sym;
      if (e & 15u) {
        sym = (int)(e >> 4);
        bitin_take(in, e & 15u);
      } else {
        sym = slow_decode(in, L.lit_sorted, L.lit_count);
        if (sym < 0) { err = INF_BAD_CODE | (5u << 8); break; }
      }
      if (sym < 256) {
        if (opos >= isize) { err = INF_OVERRUN; break; }
        if (lane == 0) out[opos] = (uint8_t)sym;
        opos++;
        continue;
      }
      if (sym == 256) break;
      sym -= 257;
      if (sym >= 29) { err = INF_BAD_CODE | (6u << 8); break; }
      uint32_t mlen;
      if (sym < 8) mlen = 3 + sym;
      else if (sym == 28) mlen = 258;
      else {
        int eb = (sym - 4) >> 2;
        mlen = 3 + ((4 + (sym & 3)) << eb) + bitin_take(in, eb);
      }
      bitin_refill(in, lane);
      uint32_t de = uni(L.dist_fast[(uint32_t)in.bb & ((1u << DIST_BITS) - 1u)]);
      int ds;
      if (de & 15u) {
        ds = (int)(de >> 4);
        bitin_take(in, de & 15u);
      } else {
        ds = slow_decode(in, L.dist_sorted, L.dist_count);
        if (ds < 0) { err = INF_BAD_CODE | (7u << 8); break; }
      }
      if (ds >= 30) { err = INF_BAD_CODE | (8u << 8); break; }
      uint32_t dist;
      if (ds < 4) dist = 1 + ds;
      else {
        int eb = (ds - 2) >> 1;
        dist = 1 + ((2 + (ds & 1)) << eb) + bitin_take(in, eb);
      }
      if (dist > opos) { err = INF_BAD_DIST; break; }
      if (opos + mlen > isize) { err = INF_OVERRUN; break; }
      if (lane == nm) { m_dst = opos; m_len = mlen; m_dist = dist; }
      nm++;
      opos += mlen;
      if (nm == WAVE) {
        resolve_batch(out, lane, nm, m_dst, m_len, m_dist);
        nm = 0;
      }
    }
    if (err) { st = (uint32_t)err; break; }
    if (bfinal) break;
  }
  if (nm && st == INF_OK) resolve_batch(out, lane, nm, m_dst, m_len, m_dist);
  if (st == INF_OK && opos != isize) st = INF_SIZE_MISMATCH;
  if (lane == 0) status[b] = st;
}

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
                                                       uint32_t* status) {
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
  const uint8_t* tr = comp + blk_coff[b + 1] - 8;
  const uint32_t want = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
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
  if (c != want && status[b] == INF_OK) status[b] = INF_CRC_MISMATCH;
}

void launch_bgzf_inflate(const uint8_t* comp, const uint64_t* blk_coff, const uint64_t* blk_uoff, uint8_t* out,
                         uint32_t n_blocks, uint32_t* status, hipStream_t st) {
  if (!n_blocks) return;
  hipLaunchKernelGGL(k_bgzf_inflate, dim3(n_blocks), dim3(WAVE), 0, st, comp, blk_coff, blk_uoff, out, n_blocks, status);
}
void launch_bgzf_crc32(const uint8_t* comp, const uint64_t* blk_coff, const uint64_t* blk_uoff, const uint8_t* out,
                       uint32_t n_blocks, uint32_t* status, hipStream_t st) {
  if (!n_blocks) return;
  hipLaunchKernelGGL(k_bgzf_crc32, dim3((n_blocks + CRC_T - 1) / CRC_T), dim3(CRC_T), 0, st, comp, blk_coff, blk_uoff, out, n_blocks, status);
}

}  // namespace bioscan
