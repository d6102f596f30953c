// inflate_v2.hip -- K1 v2: BGZF inflate with wave-parallel Huffman decoding (gfx950, wave64).
//
// Same contract as K1 v1 (inflate.hip): one BGZF member per wavefront, output window = the
// member's own range of the inflated stream.  What changes is how a DEFLATE block's symbol
// stream is decoded.  v1 walks it serially (one symbol per ~35 wave-instructions, 63 lanes
// idle).  v2 cuts the compressed body into 64 sub-streams of `sub_dw` dwords and lets every lane
// decode its own sub-stream at once:
//   1. speculative pass: lane 0 starts at the exact bit position; lane i>0 starts at its
//      sub-stream boundary (almost never a symbol start).  Huffman/DEFLATE streams
//      self-synchronise, so each lane's END position (first symbol start at/after the next
//      boundary) is usually already correct even when its start was wrong.
//   2. fix-point: lane i+1 adopts lane i's end as its start and re-decodes if that changed; repeat
//      until no lane changes (lane 0 is exact, so by induction the chain is exact -- speculation
//      only affects speed).  Lanes after the first END-OF-BLOCK are dead.
//   3. wave prefix sums of per-lane output bytes / match counts give every lane its output
//      offset; a last pass writes literals straight to the output window and appends LZ77
//      matches to the workgroup's match list (L2-resident scratch).
//   4. the match list is resolved 64 matches at a time by the dependency-ordered batch copy
//      shared with v1 (resolve_batch).
// Tables (u32 entries with length/distance base and extra-bit count folded in) live in LDS next
// to the staged compressed bytes of the round; block headers / code lengths are parsed by the
// uniform register-staged bit reader of v1.  Persistent grid: workgroups pull members from an
// atomic counter so the per-workgroup match scratch is bounded by residency.
#include "kernels.h"
#include <stdlib.h>
#include <stdio.h>

namespace bioscan {

#define WAVE 64
#ifndef V2_SUB_DW
#define V2_SUB_DW 7
#endif
#ifndef V2_GLOBAL_INPUT
#define V2_GLOBAL_INPUT 1
#endif
#ifndef V2_OV_BITS
#define V2_OV_BITS 96
#endif
#ifndef V2_WIN_BYTES
#define V2_WIN_BYTES 5632
#endif
constexpr int V2_LIT_BITS = 9;                          // zlib's root sizes: ENOUGH_LENS = 852, ENOUGH_DISTS = 592
constexpr int V2_DIST_BITS = 6;
constexpr int V2_MAX_SUB_DW = V2_SUB_DW;                // odd => conflict-free initial LDS reads
constexpr int V2_WIN = V2_WIN_BYTES;                    // LDS output window of one round (multiple of 16)
constexpr int V2_STAGE_DW = 64 * V2_MAX_SUB_DW + 8;  // only when the input is staged through LDS
constexpr int V2_LIT_SUB = 352;    // 852 - 512 = 340 sub-table entries at most
constexpr int V2_DIST_SUB = 528;   // 592 - 64
// 16-bit table entries (half the LDS of u32 entries: more resident waves), laid out so that the decode loop (VALU-issue
// bound: SQ_INSTS_VALU x 4 cycles = 96 % of K1's cycles) classifies an entry with the fewest instructions:
//   literal        len[0:3] | byte[4:11]
//   length / dist  len[0:3] | extra-bit count[4:7] | be_lut index[8:13] | E_HI   (length 257+c -> c, distance d -> 32+d)
//   end of block   a pointer to the STOP_EOB null slot (E_EOB)
//   sub-table ptr  E_SUB | entry index of the sub-table from lit_fast [4:14] | index width[0:3] (up to 9 for distances)
//   `len` of a sub-table entry excludes the root bits (consumed when the pointer is followed).
constexpr uint32_t E_SUB = 0x8000u, E_HI = 0x4000u;
// Symbols that must not occur in valid data (literal/length 286, 287, distance 30, 31) may be given code lengths by a
// header; their entries are E_BAD, so using one is an invalid code.
// A lane that has stopped parks on a null slot: an entry of zero index width that points to itself, so the lane keeps
// executing the shared instructions without changing state and without a per-lane "running" predicate.  Which slot it
// parks on says why it stopped.
constexpr uint32_t V2_NULL_BASE = 2u * ((1u << V2_LIT_BITS) + V2_LIT_SUB + (1u << V2_DIST_BITS) + V2_DIST_SUB) + 64u * 4u;  // byte offset from lit_fast
constexpr uint32_t STOP_END = V2_NULL_BASE, STOP_EOB = V2_NULL_BASE + 2u, STOP_BAD = V2_NULL_BASE + 4u;
// "no such code" is a pointer to the STOP_BAD slot: hitting it parks the lane there, the decode loop has no test for it
constexpr uint32_t E_BAD = E_SUB | ((STOP_BAD >> 1) << 4);
// End-of-block is a pointer to the STOP_EOB slot as well.  Following a pointer consumes the index width of the table
// it sits in, not the code's length: the table build records the difference in V2Lds::eob_fix and the pass subtracts
// it from the lane's end position.
constexpr uint32_t E_EOB = E_SUB | ((STOP_EOB >> 1) << 4);
constexpr uint32_t F_EOB = 1, F_BAD = 2;
// Waves of one workgroup decode different members and never exchange data: a workgroup only exists to get past
// the 16-workgroups-per-CU residency cap (K1 is latency-bound, its speed follows the number of resident waves).
// Every synchronisation is therefore wave-local: LDS operations of one wave execute in order, so a compiler +
// counter fence is all a "barrier" has to be.
#ifndef V2_WAVES_PER_WG
#define V2_WAVES_PER_WG 1
#endif
#define V2_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

// Table-build scratch (code lengths, canonical order, precode table) is only live while a block header
// is parsed, the output window only while a round is written and resolved: they share LDS.
struct V2Build {
  uint16_t lit_sorted[288];
  uint16_t dist_sorted[32];
  uint16_t lit_count[16];
  uint16_t dist_count[16];
  uint16_t t_offs[16], t_first[16], t_w[16];
  uint8_t lens[320];
  uint8_t pre_fast[128];
  uint8_t pre_lens[20];
};
struct __attribute__((aligned(16))) V2Lds {
  uint16_t lit_fast[(1 << V2_LIT_BITS) + V2_LIT_SUB];
  uint16_t dist_fast[(1 << V2_DIST_BITS) + V2_DIST_SUB];  // must follow lit_fast: the decode loop indexes both as one array
  uint32_t be_lut[64];  // [0..31] length symbols 257.., [32..63] distance symbols: base value
  uint16_t null_slot[7];  // must follow be_lut: self-pointing entries a stopped lane idles on (see v2_pass)
  uint16_t eob_fix;       // bits a lane over-consumed when it followed the end-of-block pointer (see E_EOB)
#if !V2_GLOBAL_INPUT
  uint32_t stage[V2_STAGE_DW];
#endif
#ifdef V2_PAD_LDS
  uint32_t pad_lds[V2_PAD_LDS / 4];  // occupancy experiment only
#endif
  union {
    uint8_t win[V2_WIN] __attribute__((aligned(16)));
    V2Build b;
  };
};
static_assert(sizeof(V2Build) <= V2_WIN, "output window must be able to hold the table-build scratch");

__device__ __forceinline__ uint32_t uni2(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t bitrev2(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// ---- uniform register-staged bit reader (as v1) ---------------------------------------------------
struct UBits {
  const uint32_t* base;
  uint32_t cur, nxt, cidx, wpos;
  uint64_t bb;
  int bc;
};
// start reading at bit `bitpos` counted from the 4-byte aligned pointer `base`
__device__ __forceinline__ void ub_init(UBits& s, const uint32_t* base, uint64_t bitpos, int lane) {
  s.base = base;
  uint32_t w = (uint32_t)(bitpos >> 5);
  s.cidx = w >> 6;
  s.cur = base[(size_t)s.cidx * 64 + lane];
  s.nxt = base[(size_t)(s.cidx + 1) * 64 + lane];
  uint32_t first = __builtin_amdgcn_readlane(s.cur, w & 63);
  s.wpos = w + 1;
  int skip = (int)(bitpos & 31);
  s.bb = (uint64_t)(first >> skip);
  s.bc = 32 - skip;
}
__device__ __forceinline__ uint32_t ub_next_word(UBits& s, int lane) {
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
__device__ __forceinline__ void ub_refill(UBits& s, int lane) {
  if (s.bc <= 32) {
    s.bb |= (uint64_t)ub_next_word(s, lane) << s.bc;
    s.bc += 32;
  }
}
__device__ __forceinline__ uint32_t ub_take(UBits& s, int n) {
  uint32_t v = (uint32_t)s.bb & ((1u << n) - 1u);
  s.bb >>= n;
  s.bc -= n;
  return v;
}
__device__ __forceinline__ uint64_t ub_bitpos(const UBits& s) { return (uint64_t)s.wpos * 32 - (uint64_t)s.bc; }

#ifdef V2_GUARD
__device__ unsigned int v2_guard_word[8];
#define V2_G(cond, code, val) ((cond) ? (atomicOr(&v2_guard_word[0], 1u << (code)), atomicMax(&v2_guard_word[code], (unsigned)(val)), true) : false)
#else
#define V2_G(cond, code, val) false
#endif
// ---- table entries ---------------------------------------------------------------------------------
// length symbol s = sym - 257 (0..28) / distance symbol (0..29): base value and extra-bit count (RFC 1951 3.2.5)
__device__ __forceinline__ void len_base_extra(uint32_t s, uint32_t* base, uint32_t* eb) {
  const uint32_t e = s < 8u ? 0u : (s - 4u) >> 2;
  const uint32_t b = s < 8u ? 3u + s : 3u + ((4u + (s & 3u)) << e);
  *eb = s == 28u ? 0u : e;
  *base = s == 28u ? 258u : b;
}
__device__ __forceinline__ void dist_base_extra(uint32_t s, uint32_t* base, uint32_t* eb) {
  const uint32_t e = s < 4u ? 0u : (s - 2u) >> 1;
  *eb = e;
  *base = s < 4u ? 1u + s : 1u + ((2u + (s & 1u)) << e);
}
__device__ __forceinline__ uint32_t sym_entry(int sym, int len, bool is_dist) {
  uint32_t base, eb;
  if (is_dist) {
    dist_base_extra((uint32_t)sym, &base, &eb);
    if (sym > 29) return E_BAD;
    return E_HI | ((32u + (uint32_t)sym) << 8) | (eb << 4) | (uint32_t)len;
  }
  if (sym < 256) return ((uint32_t)sym << 4) | (uint32_t)len;
  if (sym == 256) return E_EOB;
  len_base_extra((uint32_t)(sym - 257), &base, &eb);
  if (sym > 285) return E_BAD;
  return E_HI | ((uint32_t)(sym - 257) << 8) | (eb << 4) | (uint32_t)len;
}

// Build the two-level decode table of one alphabet: root table of 2^root_bits entries followed by
// sub-tables for codes longer than root_bits (canonical codes that share a root prefix are
// contiguous in (len, sym) order, so each sub-table is sized by the last = longest code of its
// group).  Returns 1 if the code is over-subscribed or the sub-table space is exhausted.
__device__ int v2_build(V2Lds& L, const uint8_t* lens, int n, uint16_t* fast, uint32_t abs_off, int root_bits, int sub_cap,
                        uint16_t* sorted, uint16_t* count, bool is_dist, int lane) {
  V2_SYNC();
  for (int i = lane; i < (1 << root_bits) + sub_cap; i += WAVE) fast[i] = (uint16_t)E_BAD;  // bit patterns no code maps to
  if (lane < 16) count[lane] = 0;
  V2_SYNC();
  // 1. histogram of code lengths: 64 symbols per step, one ballot per length value; lane L keeps count[L]
  uint32_t my_cnt = 0;
  for (int c0 = 0; c0 < n; c0 += WAVE) {
    const int sidx = c0 + lane;
    const int l = sidx < n ? (int)lens[sidx] : 0;
#pragma unroll
    for (int Lk = 1; Lk <= 15; Lk++) {
      const unsigned long long m = __ballot(l == Lk);
      if (lane == Lk) my_cnt += (uint32_t)__popcll(m);
    }
  }
  if (lane >= 1 && lane <= 15) count[lane] = (uint16_t)my_cnt;
  V2_SYNC();
  if (lane == 0) {
    uint32_t o = 0, code = 0;
    int left = 1, over = 0;
    for (int l = 1; l <= 15; l++) {
      uint32_t c = count[l];
      code <<= 1;
      L.b.t_first[l] = (uint16_t)code;
      L.b.t_offs[l] = (uint16_t)o;
      o += c;
      code += c;
      left <<= 1;
      left -= (int)c;
      if (left < 0) over = 1;
    }
    // incomplete codes: libdeflate (the inflater the reference links) accepts only an empty distance code or a code
    // with a single codeword of length 1 (build_decode_table); everything else that leaves code space unused is invalid
    if (left > 0 && !over) {
      const bool empty_ok = o == 0 && is_dist;
      const bool single_ok = o == 1 && count[1] == 1;
      if (!empty_ok && !single_ok) over = 1;
    }
    L.b.t_offs[0] = (uint16_t)o;
    L.b.t_first[0] = (uint16_t)over;
  }
  V2_SYNC();
  // 2. canonical order (by length, then symbol): rank of a symbol inside its length class = symbols of
  //    the same length with a smaller index -> per-chunk ballots with a running base per length
  {
    uint32_t run_base = (lane >= 1 && lane <= 15) ? (uint32_t)L.b.t_offs[lane] : 0u;  // lane L tracks length L
    for (int c0 = 0; c0 < n; c0 += WAVE) {
      const int sidx = c0 + lane;
      const int l = sidx < n ? (int)lens[sidx] : 0;
      uint32_t slot = 0;
#pragma unroll
      for (int Lk = 1; Lk <= 15; Lk++) {
        const unsigned long long m = __ballot(l == Lk);
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)run_base, Lk);
        if (l == Lk) slot = b + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == Lk) run_base += (uint32_t)__popcll(m);
      }
      if (l) sorted[slot] = (uint16_t)sidx;
    }
  }
  V2_SYNC();
  if (lane == 0) {
    uint32_t o = L.b.t_offs[0];
    int over = L.b.t_first[0];
    // sub-tables (serial: long codes are few)
    uint32_t k = root_bits < 15 ? L.b.t_offs[root_bits + 1] : o;
    uint32_t next_free = 1u << root_bits;
    while (k < o && !over) {
      const int sym_k = sorted[k];
      const int len_k = lens[sym_k];
      const uint32_t code_k = (uint32_t)L.b.t_first[len_k] + (k - L.b.t_offs[len_k]);
      const uint32_t prefix = code_k >> (len_k - root_bits);
      uint32_t j = k + 1;
      int max_len = len_k;
      while (j < o) {
        const int sj = sorted[j];
        const int lj = lens[sj];
        const uint32_t cj = (uint32_t)L.b.t_first[lj] + (j - L.b.t_offs[lj]);
        if ((cj >> (lj - root_bits)) != prefix) break;
        max_len = lj;
        j++;
      }
      const uint32_t sbits = (uint32_t)(max_len - root_bits);
      if (next_free + (1u << sbits) > (1u << root_bits) + (uint32_t)sub_cap) { over = 1; break; }
      fast[bitrev2(prefix, root_bits)] = (uint16_t)(E_SUB | ((abs_off + next_free) << 4) | sbits);  // sbits <= 9 (distance codes)
      for (uint32_t m = k; m < j; m++) {
        const int sm = sorted[m];
        const int lm = lens[sm];
        const uint32_t cm = (uint32_t)L.b.t_first[lm] + (m - L.b.t_offs[lm]);
        const uint32_t r = bitrev2(cm, lm) >> root_bits;  // bits after the root, LSB-first
        const uint16_t e = (uint16_t)sym_entry(sm, lm - root_bits, is_dist);  // the root bits are consumed when the pointer is followed
        if (!is_dist && sm == 256) L.eob_fix = (uint16_t)(sbits - (uint32_t)(lm - root_bits));
        for (uint32_t i = r; i < (1u << sbits); i += (1u << (lm - root_bits))) fast[next_free + i] = e;
      }
      next_free += 1u << sbits;
      k = j;
    }
    L.b.t_first[0] = (uint16_t)over;
  }
  V2_SYNC();
  if (uni2(L.b.t_first[0])) return 1;
  const uint32_t o = uni2(L.b.t_offs[0]);
  for (uint32_t k = lane; k < o; k += WAVE) {
    int sym = sorted[k];
    int l = lens[sym];
    if (l <= root_bits) {
      uint32_t c = (uint32_t)L.b.t_first[l] + (k - L.b.t_offs[l]);
      uint32_t r = bitrev2(c, l);
      const uint16_t e = (uint16_t)sym_entry(sym, l, is_dist);
      if (!is_dist && sym == 256) L.eob_fix = (uint16_t)(root_bits - l);
      for (uint32_t i = r; i < (1u << root_bits); i += (1u << l)) fast[i] = e;
    }
  }
  V2_SYNC();
  return 0;
}

// One decode pass of this lane's sub-stream [start, limit).  MODE 0: count only; 1: write literals to
// HBM; 2: write literals to the LDS window.  The lane keeps a 64-bit bit buffer in registers and
// refills it one dword at a time from the staged input.  The body is written predicated (selects,
// wave-uniform ballot branches, no break) so it compiles to straight-line code instead of nested
// exec-mask regions.
template <int MODE, bool SPEC = false>
__device__ __forceinline__ void v2_pass(V2Lds& L, bool active, uint32_t start, uint32_t limit, uint32_t& end_out,
                                        uint32_t& nout, uint32_t& nmatch, uint32_t& flags, uint8_t* out, uint32_t opos,
                                        unsigned long long* mlist, uint32_t mpos, uint32_t win_base,
                                        uint32_t count_from, uint32_t& first_out, const uint32_t* __restrict__ gsrc) {
  constexpr bool WRITE = MODE != 0;
  static_assert(offsetof(V2Lds, null_slot) - offsetof(V2Lds, lit_fast) == V2_NULL_BASE, "null slots must sit at V2_NULL_BASE");
  uint32_t pos = start;
  uint32_t acc = 0;  // MODE 0: bytes produced [0:19] | matches [20:31] since `first`
  const bool run0 = active && pos < limit;
  // speculative lanes start `overlap` bits early: symbols that begin before count_from only serve to
  // synchronise; the first symbol start at/after count_from is reported and counting restarts there.
  // first == ~0 means "not reached yet".
  uint32_t first = (!SPEC || pos >= count_from) ? pos : 0xFFFFFFFFu;
#if V2_GLOBAL_INPUT
#ifdef V2_GUARD
#define V2_SRC(i) (V2_G((i) > ((limit + 64u) >> 5) + 3u, 1, (i)) ? 0u : gsrc[i])
#else
#define V2_SRC(i) gsrc[i]
#endif
#else
#define V2_SRC(i) L.stage[i]
#endif
  // The lane's bit window is 32 bits starting at `pos`, funnel-shifted out of two input dwords d0 (dword wp) and d1;
  // `nxt` is dword wp + 2, prefetched.  A symbol consumes <= 28 bits, so pos crosses at most one dword per step.
  // (a lane that does not run keeps pos, so it never crosses and never loads again: its three reads are parked at 0)
  uint32_t wp = pos >> 5;
  const uint32_t wp0 = run0 ? wp : 0u;
  uint32_t d0 = V2_SRC(wp0), d1 = V2_SRC(wp0 + 1), nxt = V2_SRC(wp0 + 2);
  // One table lookup per iteration.  A lane is a small state machine: `tb/mb` describe its next lookup (byte offset
  // of the table from lit_fast, index width).  tb == 0 is the literal/length root (a symbol boundary), tb == DIST_BASE
  // the distance root of a pending match, a value below the null slots a sub-table (below DIST_BASE: literal/length),
  // a null slot a stopped lane.  A sub-table pointer consumes the root bits and re-targets the next lookup, a length
  // symbol switches the lane to the distance table, a stopped lane follows its self-pointer for ever: lanes in
  // different states share the same instructions, so a wave never pays for a path only one lane needs, and there is
  // no loop-carried predicate (each costs four scalar instructions per step to merge; K1 is bound by VALU + SALU issue).
  const uint8_t* __restrict__ T = (const uint8_t*)L.lit_fast;  // dist_fast follows lit_fast in LDS
  constexpr uint32_t DIST_BASE = 2u * ((1u << V2_LIT_BITS) + V2_LIT_SUB);
  uint32_t tb = run0 ? 0u : STOP_END, mb = run0 ? (uint32_t)V2_LIT_BITS : 0u, mlen = 0;
  while (__ballot(tb < V2_NULL_BASE) != 0ull) {
#ifdef V2_ASM_MARKERS
    asm volatile("; V2LOOP_BEGIN %0" ::"n"(MODE));
#endif
    const bool in_lit = tb < DIST_BASE;
    const uint32_t w = __builtin_amdgcn_alignbit(d1, d0, pos & 31u);
    // the load is written out: selected by the compiler, the 16-bit LDS read is followed by an `and 0xffff` the
    // hardware has already done (ds_read_u16 zero-extends)
    uint32_t e;
    {
      const uint32_t ea = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)T + tb + (__builtin_amdgcn_ubfe(w, 0u, mb) << 1);
      asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(ea));
    }
    __builtin_assume(e < 65536u);
    const uint32_t l = e & 15u;                            // code length (pointer: index width of the sub-table)
    const bool ptr = e >= E_SUB;                           // pointer to a second-level table (or a null slot)
    const bool is_lit = e < 0x1000u;                       // literal entries are 0x001 .. 0xFFF
    // length / distance entry: no entry lies in 0x1000 .. 0x3FFF, so it is "neither of the two" -- computed on the lane
    // masks (one scalar op); written as a lane predicate the compiler turns it back into a vector range test
    const bool lenlike = __builtin_amdgcn_inverse_ballot_w64(~(__ballot(is_lit) | __ballot(ptr)));
    const bool is_len = lenlike && in_lit;
    const bool is_dist = lenlike && !is_len;
    const uint32_t base = L.be_lut[(e >> 8) & 63u];
    const uint32_t ebv = lenlike ? ((e >> 4) & 15u) : 0u;
    const uint32_t val = base + __builtin_amdgcn_ubfe(w, l, ebv);
    pos += ptr ? mb : l + ebv;
    if ((pos >> 5) != wp) {
      // explicit moves: left to the register allocator, the fresh load is copied into place right away and the
      // wave waits for it here instead of one crossing later
      asm volatile("v_mov_b32 %0, %1" : "=v"(d0) : "v"(d1));
      asm volatile("v_mov_b32 %0, %1" : "=v"(d1) : "v"(nxt));
      wp++;
      nxt = V2_SRC(wp + 2);
    }
    bool bad = false;                                      // (an unassigned code is a pointer to STOP_BAD)
    if (MODE == 1) { if (is_lit && !V2_G(opos >= win_base, 2, opos)) out[opos] = (uint8_t)(e >> 4); }
    if (MODE == 2) { if (is_lit) (L.win - win_base)[opos] = (uint8_t)(e >> 4); }  // base pointer folded: one VALU less than an index subtraction
    if (is_len) mlen = WRITE ? val : val + (1u << 20);   // count passes carry the match count in the same accumulator
    bool okm = is_dist;
    if (WRITE) {
      if (okm && val > opos) { bad = true; okm = false; }
      if (okm && !V2_G(mpos >= V2_SCRATCH_STRIDE, 3, mpos)) { uint2 ent; ent.x = opos; ent.y = mlen | (val << 12); ((uint2*)mlist)[mpos] = ent; }  // = opos | mlen << 32 | val << 44
      mpos += okm ? 1u : 0u;
    }
    const uint32_t produced = is_lit ? 1u : (okm ? mlen : 0u);
    if (WRITE) opos += produced; else acc += produced;
    // next lookup: a completed symbol returns to the literal/length root, or parks if the sub-stream is used up
    const bool at_end = pos >= limit;
    uint32_t ntb = is_len ? DIST_BASE : (at_end ? STOP_END : 0u);
    uint32_t nmb = is_len ? (uint32_t)V2_DIST_BITS : (at_end ? 0u : (uint32_t)V2_LIT_BITS);
    if (ptr) { ntb = (e >> 3) & 0xFFEu; nmb = l; }
    if (bad) { ntb = STOP_BAD; nmb = 0u; }
    tb = ntb; mb = nmb;
    if (SPEC) {
      // the next step starts a symbol at / after count_from: counting restarts there
      const bool cross = tb == 0u && first == 0xFFFFFFFFu && pos >= count_from;
      if (cross) { first = pos; acc = 0; }
    }
#ifdef V2_ASM_MARKERS
    asm volatile("; V2LOOP_END %0" ::"n"(MODE));
#endif
  }
  if (active) {
    // a lane that stopped before reaching count_from (bogus EOB / bad code while synchronising) has no valid result,
    // even if it stopped exactly on a true symbol boundary: first stays ~0, which never equals a predecessor's end
    // and so forces a re-decode
    if (first == 0xFFFFFFFFu) acc = 0;
    end_out = tb == STOP_EOB ? pos - (uint32_t)L.eob_fix : pos; nout = acc & 0xFFFFFu; nmatch = acc >> 20; first_out = first;
    flags = tb == STOP_EOB ? F_EOB : (tb == STOP_BAD ? F_BAD : 0u);
  }
}

__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, int lane, uint32_t* total) {
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    uint32_t o = __shfl_up(inc, d, WAVE);
    if (lane >= d) inc += o;
  }
  *total = __builtin_amdgcn_readlane(inc, 63);
  return inc - v;
}

// unaligned vector access helpers (gfx950 runs with unaligned global access enabled)
typedef uint32_t u32x4_raw __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) u32x4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) u64p { uint64_t v; };
struct __attribute__((packed, aligned(1))) u32p { uint32_t v; };
struct __attribute__((packed, aligned(1))) u16p { uint16_t v; };
__device__ __forceinline__ u32x4 ld16(const uint8_t* p) { return *(const u32x4*)p; }
__device__ __forceinline__ void st16(uint8_t* p, u32x4 v) { *(u32x4*)p = v; }
__device__ __forceinline__ void st8(uint8_t* p, uint64_t v) { ((u64p*)p)->v = v; }
__device__ __forceinline__ void st4(uint8_t* p, uint32_t v) { ((u32p*)p)->v = v; }
__device__ __forceinline__ void st2(uint8_t* p, uint16_t v) { ((u16p*)p)->v = v; }

// dependency-ordered copy of <= 64 matches (one per lane)
__device__ void v2_resolve_batch(uint8_t* out, int lane, int nm, uint32_t m_dst, uint32_t m_len, uint32_t m_dist) {
  bool valid = lane < nm;
  if (valid && V2_G(m_dst + m_len > 65536u || m_dist > m_dst || m_len > 258u, 4, m_dst + m_len)) valid = false;
  const uint32_t src_lo = m_dst - m_dist;
  const uint32_t src_end = src_lo + m_len;
  const uint32_t src_hi = src_end < m_dst ? src_end : m_dst;
  const uint32_t first_dst = __builtin_amdgcn_readlane(m_dst, 0);
  uint64_t dep = 0;
  const bool maybe = valid && src_hi > first_dst;
  if (__ballot(maybe) != 0ull) {
    // destinations are sorted and disjoint: the earlier matches overlapping [src_lo, src_hi) are the
    // index range [first i with dst_end_i > src_lo, last i with dst_i < src_hi]; two binary searches
    // over the lanes (ds_bpermute) instead of a 63-step sweep.
    const uint32_t dend = valid ? m_dst + m_len : 0xFFFFFFFFu;
    const uint32_t dbeg = valid ? m_dst : 0xFFFFFFFFu;
    int lo1 = 0, hi1 = nm, lo2 = 0, hi2 = nm;
#pragma unroll
    for (int step = 0; step < 7; step++) {
      const int mid1 = (lo1 + hi1) >> 1, mid2 = (lo2 + hi2) >> 1;
      const uint32_t v1 = (uint32_t)__shfl((int)dend, mid1 & 63, WAVE);
      const uint32_t v2 = (uint32_t)__shfl((int)dbeg, mid2 & 63, WAVE);
      if (lo1 < hi1) { if (v1 > src_lo) hi1 = mid1; else lo1 = mid1 + 1; }
      if (lo2 < hi2) { if (v2 >= src_hi) hi2 = mid2; else lo2 = mid2 + 1; }
    }
    // lo1 = first overlapping index, lo2 = count of matches with dst < src_hi
    int a = lo1, b = lo2 - 1;
    if (b > lane - 1) b = lane - 1;
    if (maybe && a <= b) {
      const uint64_t hi_mask = b >= 63 ? ~0ull : ((1ull << (b + 1)) - 1ull);
      dep = hi_mask & ~((1ull << a) - 1ull);
    }
  }
  const uint64_t all = nm >= 64 ? ~0ull : ((1ull << nm) - 1ull);
  uint64_t done = 0;
  while (done != all) {
    const bool ready = valid && !((done >> lane) & 1ull) && ((dep & ~done) == 0ull);
    if (ready) {
      uint8_t* d = out + m_dst;
      const uint8_t* s = out + src_lo;
      if (m_dist >= 16) {
        // source and destination are >= 16 bytes apart: stream 16-byte unaligned vectors
        uint32_t k = 0;
        for (; k + 16 <= m_len; k += 16) st16(d + k, ld16(s + k));
        const uint32_t rem = m_len - k;
        if (rem) {
          const u32x4 v = ld16(s + k);  // over-read is inside the (padded) buffer
          uint8_t* t = d + k;
          uint32_t o = 0;
          if (rem & 8) { st8(t, (uint64_t)v.x | ((uint64_t)v.y << 32)); o = 8; }
          if (rem & 4) { st4(t + o, o ? v.z : v.x); o += 4; }
          // remaining 0..3 bytes come from dword (o/4) of v
          const uint32_t w = o == 0 ? v.x : o == 4 ? v.y : o == 8 ? v.z : v.w;
          if (rem & 2) { st2(t + o, (uint16_t)w); if (rem & 1) t[o + 2] = (uint8_t)(w >> 16); }
          else if (rem & 1) t[o] = (uint8_t)w;
        }
      } else if (m_dist >= 4) {
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    done |= __ballot(ready);
  }
}

// Resolve of a round whose output lives in the LDS window `win` (absolute output position R maps to win[0]).  Sources
// before R are final bytes in HBM, sources at or after R are in the window.  58 % of the matches of a BAM member read
// only bytes before R: they need no ordering at all, so the list is walked twice -- first every match copies the part
// of its source that precedes R (v2_far_copy) and the matches that also read the window are compacted to the front
// of the list, then only those go through the dependency-ordered copy (v2_near_batch), in dense batches of 64.
__device__ __forceinline__ void v2_far_copy(uint8_t* win, const uint8_t* out, uint32_t R, bool valid, uint32_t m_dst, uint32_t m_len,
                                            uint32_t m_dist) {
  const uint32_t src_lo = m_dst - m_dist;
  uint32_t n_far = 0;
  if (valid && src_lo < R) { n_far = R - src_lo; if (n_far > m_len) n_far = m_len; }
  if (n_far) {
    // LDS takes unaligned 4 / 8-byte stores on gfx950: the bytes go out in the widest pieces that fit
    uint8_t* d = win + (m_dst - R);
    const uint8_t* s = out + src_lo;
    uint32_t k = 0;
    for (; k + 16 <= n_far; k += 16) {
      const u32x4 v = ld16(s + k);
      st8(d + k, (uint64_t)v.x | ((uint64_t)v.y << 32));
      st8(d + k + 8, (uint64_t)v.z | ((uint64_t)v.w << 32));
    }
    const uint32_t rem = n_far - k;
    if (rem) {
      const u32x4 v = ld16(s + k);  // over-read is inside the (padded) buffer
      uint8_t* t = d + k;
      uint32_t o = 0;
      if (rem & 8) { st8(t, (uint64_t)v.x | ((uint64_t)v.y << 32)); o = 8; }
      if (rem & 4) { st4(t + o, o ? v.z : v.x); o += 4; }
      const uint32_t w = o == 0 ? v.x : o == 4 ? v.y : o == 8 ? v.z : v.w;
      if (rem & 2) { st2(t + o, (uint16_t)w); if (rem & 1) t[o + 2] = (uint8_t)(w >> 16); }
      else if (rem & 1) t[o] = (uint8_t)w;
    }
  }
}

// dependency-ordered copy of the in-window part of <= 64 matches (one per lane, sorted by destination)
__device__ void v2_near_batch(uint8_t* win, uint32_t R, int lane, int nm, uint32_t m_dst, uint32_t m_len, uint32_t m_dist) {
  bool valid = lane < nm;
  if (valid && V2_G(m_dst + m_len > 65536u || m_dist > m_dst || m_len > 258u || m_dst < R || m_dst + m_len - R > (uint32_t)V2_WIN, 5, m_dst + m_len)) valid = false;
  const uint32_t src_lo = m_dst - m_dist;
  const uint32_t src_end = src_lo + m_len;
  const uint32_t src_hi = src_end < m_dst ? src_end : m_dst;
  const uint32_t first_dst = __builtin_amdgcn_readlane(m_dst, 0);
  uint64_t dep = 0;
  const bool maybe = valid && src_hi > first_dst;
  if (__ballot(maybe) != 0ull) {
    // destinations are sorted and disjoint: the earlier matches overlapping [src_lo, src_hi) are the
    // index range [first i with dst_end_i > src_lo, last i with dst_i < src_hi]; two binary searches
    // over the lanes (ds_bpermute) instead of a 63-step sweep.
    const uint32_t dend = valid ? m_dst + m_len : 0xFFFFFFFFu;
    const uint32_t dbeg = valid ? m_dst : 0xFFFFFFFFu;
    int lo1 = 0, hi1 = nm, lo2 = 0, hi2 = nm;
#pragma unroll
    for (int step = 0; step < 7; step++) {
      const int mid1 = (lo1 + hi1) >> 1, mid2 = (lo2 + hi2) >> 1;
      const uint32_t v1 = (uint32_t)__shfl((int)dend, mid1 & 63, WAVE);
      const uint32_t v2 = (uint32_t)__shfl((int)dbeg, mid2 & 63, WAVE);
      if (lo1 < hi1) { if (v1 > src_lo) hi1 = mid1; else lo1 = mid1 + 1; }
      if (lo2 < hi2) { if (v2 >= src_hi) hi2 = mid2; else lo2 = mid2 + 1; }
    }
    int a = lo1, b = lo2 - 1;
    if (b > lane - 1) b = lane - 1;
    if (maybe && a <= b) {
      const uint64_t hi_mask = b >= 63 ? ~0ull : ((1ull << (b + 1)) - 1ull);
      dep = hi_mask & ~((1ull << a) - 1ull);
    }
  }
  uint32_t n_far = 0;  // already copied by v2_far_copy
  if (valid && src_lo < R) { n_far = R - src_lo; if (n_far > m_len) n_far = m_len; }
  const uint64_t all = nm >= 64 ? ~0ull : ((1ull << nm) - 1ull);
  uint64_t done = 0;
  while (done != all) {
    const bool ready = valid && !((done >> lane) & 1ull) && ((dep & ~done) == 0ull);
    if (ready && n_far < m_len) {
      uint8_t* d = win + (m_dst - R);
      const uint8_t* s = win + (src_lo - R);  // only indexed at k >= n_far, where src_lo + k >= R
      uint32_t k = n_far;
      if (m_dist >= 8) {
        for (; k + 8 <= m_len; k += 8) st8(d + k, ((const u64p*)(s + k))->v);
      }
      if (m_dist >= 4) {
        for (; k + 4 <= m_len; k += 4) st4(d + k, ((const u32p*)(s + k))->v);
      }
      for (; k < m_len; k++) d[k] = s[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    done |= __ballot(ready);
  }
}

#ifndef V2_WAVES_PER_EU
#define V2_WAVES_PER_EU 5
#endif
__global__ __launch_bounds__(WAVE * V2_WAVES_PER_WG, V2_WAVES_PER_EU) void k_bgzf_inflate_v2(const uint8_t* __restrict__ comp,
                                                           const uint64_t* __restrict__ blk_coff,
                                                           const uint64_t* __restrict__ blk_uoff, uint8_t* out_all,
                                                           uint32_t n_blocks, uint32_t* __restrict__ status,
                                                           uint32_t* counter, unsigned long long* scratch,
                                                           uint32_t scratch_stride, uint32_t* dbg, uint32_t ablate, uint32_t dbg_block) {
  __shared__ V2Lds L_all[V2_WAVES_PER_WG];
  V2Lds& L = L_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  unsigned long long* mlist = scratch + ((size_t)blockIdx.x * V2_WAVES_PER_WG + (threadIdx.x >> 6)) * scratch_stride;
  uint32_t dbg_rounds = 0, dbg_passes = 0, dbg_matches = 0, dbg_near = 0;
#ifdef V2_FIXSTAT
  uint32_t fs_lanes[4] = {0, 0, 0, 0}, fs_iters[4] = {0, 0, 0, 0};  // lanes re-decoded by / runs of the 1st, 2nd, 3rd, later fix pass
#endif
  unsigned long long tc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0 = 0;
#define TICK() (t0 = dbg ? clock64() : 0)
#define TOCK(i) do { if (dbg) { unsigned long long t1 = clock64(); tc[i] += t1 - t0; t0 = t1; } } while (0)

  // base / extra-bit LUT of the length and distance symbols (RFC 1951 3.2.5), once per wave
  {
    uint32_t base, eb;
    if (lane < 32) { len_base_extra((uint32_t)lane, &base, &eb); L.be_lut[lane] = lane > 28 ? 0u : base; }
    else { dist_base_extra((uint32_t)lane - 32u, &base, &eb); L.be_lut[lane] = lane - 32 > 29 ? 0u : base; }
    if (lane < 7) L.null_slot[lane] = (uint16_t)(E_SUB | (((V2_NULL_BASE >> 1) + (uint32_t)lane) << 4));
  }
  V2_SYNC();

  for (;;) {
    uint32_t b = 0;
    if (lane == 0) b = atomicAdd(counter, 1u);
    b = uni2(b);
    if (b >= n_blocks) break;

    const uint64_t coff = blk_coff[b];
    const uint64_t cend = blk_coff[b + 1];
    const uint8_t* hdr = comp + coff;
    uint8_t* out = out_all + blk_uoff[b];
    const uint32_t isize = (uint32_t)(blk_uoff[b + 1] - blk_uoff[b]);
    uint32_t st = INF_OK;
    const uint32_t xlen = uni2((uint32_t)hdr[10] | ((uint32_t)hdr[11] << 8));
    const uint32_t magic = uni2((uint32_t)hdr[0] | ((uint32_t)hdr[1] << 8) | ((uint32_t)hdr[2] << 16) | ((uint32_t)hdr[3] << 24));
    if ((magic & 0x04FFFFFFu) != 0x04088B1Fu) {
      if (lane == 0) status[b] = INF_BAD_HEADER;
      continue;
    }
    const uint8_t* payload = hdr + 12 + xlen;
    const uint64_t payload_len = (cend - coff) - 12 - xlen - 8;
    // all bit positions are counted from the 4-byte aligned word at/before the payload
    const uint32_t* base32 = (const uint32_t*)((uintptr_t)payload & ~(uintptr_t)3);
    const uint64_t skew = (uint64_t)((uintptr_t)payload & 3) * 8;
    const uint64_t end_bits = skew + payload_len * 8;
    uint64_t P = skew;
    uint32_t opos = 0;

    for (;;) {  // DEFLATE blocks
      TICK();
      UBits in;
      ub_init(in, base32, P, lane);
      ub_refill(in, lane);
      const uint32_t bfinal = ub_take(in, 1);
      const uint32_t btype = ub_take(in, 2);
      if (btype == 3) { st = INF_BAD_BTYPE; break; }
      if (btype == 0) {
        ub_take(in, in.bc & 7);
        const uint64_t bytepos = (ub_bitpos(in) - skew) >> 3;  // relative to payload
        const uint8_t* p = payload + bytepos;
        const uint32_t len = uni2((uint32_t)p[0] | ((uint32_t)p[1] << 8));
        const uint32_t nlen = uni2((uint32_t)p[2] | ((uint32_t)p[3] << 8));
        if ((len ^ 0xFFFFu) != nlen) { st = INF_BAD_STORED; break; }
        if (opos + len > isize || bytepos + 4 + len > payload_len) { st = INF_OVERRUN; break; }
        p += 4;
        for (uint32_t k = lane; k < len; k += WAVE) out[opos + k] = p[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        opos += len;
        P = skew + (bytepos + 4 + len) * 8;
        if (bfinal) break;
        continue;
      }
      if (btype == 1) {
        for (int i = lane; i < 320; i += WAVE) {
          uint8_t l;
          if (i < 144) l = 8; else if (i < 256) l = 9; else if (i < 280) l = 7; else if (i < 288) l = 8; else l = 5;
          L.b.lens[i] = l;
        }
        V2_SYNC();
      } else {
        ub_refill(in, lane);
        const uint32_t hlit = ub_take(in, 5) + 257;
        const uint32_t hdist = ub_take(in, 5) + 1;
        const uint32_t hclen = ub_take(in, 4) + 4;
        if (hlit > 286 || hdist > 30) { st = INF_BAD_CODE | (1u << 8); break; }
        if (lane < 20) L.b.pre_lens[lane] = 0;
        V2_SYNC();
        {
          const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
          for (uint32_t i = 0; i < hclen; i++) {
            ub_refill(in, lane);
            uint32_t v = ub_take(in, 3);
            if (lane == 0) L.b.pre_lens[order[i]] = (uint8_t)v;
          }
        }
        V2_SYNC();
        for (int i = lane; i < 128; i += WAVE) L.b.pre_fast[i] = 0;
        V2_SYNC();
        if (lane == 0) {
          uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          for (int s = 0; s < 19; s++) cnt[L.b.pre_lens[s]]++;
          cnt[0] = 0;
          uint32_t next[8];
          uint32_t code = 0;
          for (int l = 1; l <= 7; l++) { code = (code + cnt[l - 1]) << 1; next[l] = code; }
          // code space of the code-length code: over-subscribed is invalid, incomplete only passes as a single 1-bit code
          uint32_t used = 0, nsym = 0;
          for (int l = 1; l <= 7; l++) { used += cnt[l] << (7 - l); nsym += cnt[l]; }
          if (used > 128u || (used < 128u && !(nsym == 1 && cnt[1] == 1))) L.b.pre_lens[19] = 1;
          for (int s = 0; s < 19; s++) {
            int l = L.b.pre_lens[s];
            if (!l) continue;
            uint32_t r = bitrev2(next[l]++, l);
            for (uint32_t i = r; i < 128; i += (1u << l)) L.b.pre_fast[i] = (uint8_t)((s << 3) | l);
          }
        }
        V2_SYNC();
        {
          const uint32_t total = hlit + hdist;
          uint32_t i = 0, prev = 0;
          int bad = (int)uni2((uint32_t)L.b.pre_lens[19]);  // invalid code-length code (set by the table build above)
          while (i < total && !bad) {
            ub_refill(in, lane);
            uint32_t e = uni2(L.b.pre_fast[(uint32_t)in.bb & 127]);
            uint32_t l = e & 7, sym = e >> 3;
            if (l == 0) { bad = 1; break; }
            ub_take(in, l);
            if (sym < 16) {
              if (lane == 0) L.b.lens[i < hlit ? i : 288 + (i - hlit)] = (uint8_t)sym;
              prev = sym;
              i++;
            } else {
              uint32_t rep, val;
              if (sym == 16) { if (i == 0) { bad = 1; break; } rep = 3 + ub_take(in, 2); val = prev; }
              else if (sym == 17) { rep = 3 + ub_take(in, 3); val = 0; }
              else { rep = 11 + ub_take(in, 7); val = 0; }
              if (i + rep > total) { bad = 1; break; }
              if (lane == 0)
                for (uint32_t k = 0; k < rep; k++) {
                  uint32_t j = i + k;
                  L.b.lens[j < hlit ? j : 288 + (j - hlit)] = (uint8_t)val;
                }
              if (sym != 16) prev = 0;
              i += rep;
            }
          }
          if (bad) { st = INF_BAD_CODE | (2u << 8); break; }
          for (uint32_t k = hlit + lane; k < 288; k += WAVE) L.b.lens[k] = 0;
          for (uint32_t k = 288 + hdist + lane; k < 320; k += WAVE) L.b.lens[k] = 0;
          V2_SYNC();
        }
      }
      if (v2_build(L, L.b.lens, 288, L.lit_fast, 0u, V2_LIT_BITS, V2_LIT_SUB, L.b.lit_sorted, L.b.lit_count, false, lane)) { st = INF_BAD_CODE | (3u << 8); break; }
      if (v2_build(L, L.b.lens + 288, 32, L.dist_fast, (1u << V2_LIT_BITS) + V2_LIT_SUB, V2_DIST_BITS, V2_DIST_SUB, L.b.dist_sorted, L.b.dist_count, true, lane)) { st = INF_BAD_CODE | (4u << 8); break; }
      P = ub_bitpos(in);
      TOCK(0);

      // ---- rounds over the block body ----
      bool block_done = false;
      while (!block_done) {
        // sub-stream size: cover what is left of the payload with 64 lanes, 5..17 dwords (odd)
        uint64_t rem_bits = end_bits > P ? end_bits - P : 0;
        uint32_t sub_dw = (uint32_t)((rem_bits + 64ull * 32 - 1) / (64ull * 32));
        if (sub_dw > (uint32_t)V2_MAX_SUB_DW) sub_dw = V2_MAX_SUB_DW;
        if (sub_dw < 5) sub_dw = 5;
        sub_dw |= 1u;
        const uint32_t subb = sub_dw * 32;
        const uint64_t wb = P >> 5;
        const uint32_t nstage = 64 * sub_dw + 6;
        V2_SYNC();
#if !V2_GLOBAL_INPUT
        for (uint32_t k = lane; k < nstage; k += WAVE) L.stage[k] = base32[wb + k];
#else
        (void)nstage;
#endif
        V2_SYNC();
        TOCK(1);
        const uint32_t rel0 = (uint32_t)(P & 31);
        const uint32_t bnd = rel0 + (uint32_t)lane * subb;
        const uint32_t limit = rel0 + (uint32_t)(lane + 1) * subb;
        uint32_t start = bnd, end = bnd, nout = 0, nmatch = 0, flags = 0;
        {
          // lanes too close to the round's start for a full pre-roll begin at the round's exact first bit instead
          const uint32_t room = (uint32_t)lane * subb;
          const uint32_t ov = room < (uint32_t)V2_OV_BITS ? room : (uint32_t)V2_OV_BITS;
          uint32_t first = bnd;
          v2_pass<0, true>(L, true, bnd - ov, limit, end, nout, nmatch, flags, nullptr, 0, nullptr, 0, 0, bnd, first, base32 + wb);
          start = first;  // counts are valid from here
        }
        dbg_passes++;
        for (int it = 0; it < 66; it++) {
          const unsigned long long stopm = __ballot(flags != 0);
          const int first_stop = stopm ? __builtin_ctzll(stopm) : 64;
          uint32_t pe = __shfl_up(end, 1, WAVE);
          const bool alive = lane > 0 && lane <= first_stop;
          const bool changed = alive && pe != start;
          if (__ballot(changed) == 0ull) break;
#ifdef V2_FIXSTAT
          { const int k = it < 3 ? it : 3; fs_lanes[k] += (uint32_t)__popcll(__ballot(changed)); fs_iters[k]++; }
#endif
          if (changed) start = pe;
          // a lane whose corrected start already lies beyond its limit owns no symbols
          if (changed && start >= limit) { end = start; nout = 0; nmatch = 0; flags = 0; }
          { uint32_t f_ = 0; v2_pass<0>(L, changed && start < limit, start, limit, end, nout, nmatch, flags, nullptr, 0, nullptr, 0, 0, 0, f_, base32 + wb); }
          dbg_passes++;
        }
        TOCK(2);
        const unsigned long long stopm = __ballot(flags != 0);
        const int last = stopm ? __builtin_ctzll(stopm) : 63;
        const uint32_t last_flags = __builtin_amdgcn_readlane(flags, last);
        if (stopm && (last_flags & F_BAD)) { st = INF_BAD_CODE | (5u << 8); break; }
        const bool valid = lane <= last;
        uint32_t tot_out, tot_m;
        const uint32_t obase = opos + wave_excl_scan_u32(valid ? nout : 0u, lane, &tot_out);
        const uint32_t mbase = wave_excl_scan_u32(valid ? nmatch : 0u, lane, &tot_m);
        if (opos + tot_out > isize) { st = INF_OVERRUN; break; }
        if (tot_m > scratch_stride) { st = INF_OVERRUN | (1u << 8); break; }
        // write pass: literals to the (LDS or HBM) window, matches to the list
        const bool use_win = tot_out <= (uint32_t)V2_WIN;
        {
          uint32_t e2 = 0, o2 = 0, m2 = 0, f2 = 0;
          if (!(ablate & 2u)) {
            uint32_t f_ = 0;
            if (use_win) v2_pass<2>(L, valid && start < limit, start, limit, e2, o2, m2, f2, out, obase, mlist, mbase, opos, 0, f_, base32 + wb);
#ifdef V2_GUARD
            else v2_pass<1>(L, valid && start < limit, start, limit, e2, o2, m2, f2, out, obase, mlist, mbase, isize, 0, f_, base32 + wb);
#else
            else v2_pass<1>(L, valid && start < limit, start, limit, e2, o2, m2, f2, out, obase, mlist, mbase, opos, 0, f_, base32 + wb);
#endif
          }
          dbg_passes++;
          if (__ballot(valid && (f2 & F_BAD)) != 0ull) { st = INF_BAD_DIST; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        TOCK(3);
        if (!(ablate & 1u)) {
          unsigned long long m_next = 0;
          if ((uint32_t)lane < tot_m) m_next = mlist[lane];
          if (use_win) {
            uint32_t n_near = 0;  // matches that also read this round's window, compacted to the front of the list
            for (uint32_t k = 0; k < tot_m; k += WAVE) {
              const uint32_t nmb = tot_m - k < WAVE ? tot_m - k : WAVE;
              const unsigned long long m = m_next;
              if (k + WAVE + (uint32_t)lane < tot_m) m_next = mlist[k + WAVE + lane];  // prefetch the next batch
              const uint32_t md = (uint32_t)(m & 0xFFFFFFFFull), ml = (uint32_t)((m >> 32) & 0xFFFu), mdist = (uint32_t)(m >> 44);
              const bool valid_m = (uint32_t)lane < nmb;
              v2_far_copy(L.win, out, opos, valid_m, md, ml, mdist);
              const bool near = valid_m && md - mdist + ml > opos;
              const unsigned long long nmask = __ballot(near);
              if (near) mlist[n_near + __builtin_amdgcn_mbcnt_hi((uint32_t)(nmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nmask, 0u))] = m;  // < k + 64: never a slot still to be read
              n_near += (uint32_t)__popcll(nmask);
              if (dbg) { dbg_matches += nmb; dbg_near += (uint32_t)__popcll(nmask); }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if ((uint32_t)lane < n_near) m_next = mlist[lane];
            for (uint32_t k = 0; k < n_near; k += WAVE) {
              const uint32_t nmb = n_near - k < WAVE ? n_near - k : WAVE;
              const unsigned long long m = m_next;
              if (k + WAVE + (uint32_t)lane < n_near) m_next = mlist[k + WAVE + lane];
              const uint32_t md = (uint32_t)(m & 0xFFFFFFFFull), ml = (uint32_t)((m >> 32) & 0xFFFu), mdist = (uint32_t)(m >> 44);
              v2_near_batch(L.win, opos, lane, (int)nmb, md, ml, mdist);
            }
          } else {
            for (uint32_t k = 0; k < tot_m; k += WAVE) {
              const uint32_t nmb = tot_m - k < WAVE ? tot_m - k : WAVE;
              const unsigned long long m = m_next;
              if (k + WAVE + (uint32_t)lane < tot_m) m_next = mlist[k + WAVE + lane];  // prefetch the next batch
              const uint32_t md = (uint32_t)(m & 0xFFFFFFFFull), ml = (uint32_t)((m >> 32) & 0xFFFu), mdist = (uint32_t)(m >> 44);
              v2_resolve_batch(out, lane, (int)nmb, md, ml, mdist);
            }
          }
          if (use_win) {
            // coalesced flush of the window (16 B per lane; the destination may be unaligned)
            uint8_t* dstp = out + opos;
            const uint32_t full = tot_out & ~15u;
            for (uint32_t i = (uint32_t)lane * 16; i < full; i += WAVE * 16) {
              const uint32_t* w = (const uint32_t*)(L.win + i);
              u32x4 v;
              v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
              st16(dstp + i, v);
            }
            if ((uint32_t)lane < (tot_out & 15u)) dstp[full + lane] = L.win[full + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          }
        }
        TOCK(4);
        opos += tot_out;
        dbg_rounds++;
        const uint32_t end_last = __builtin_amdgcn_readlane(end, last);
        P = (wb << 5) + end_last;
        if (stopm) block_done = true;
        else if (P >= end_bits + 64) { st = INF_OVERRUN | (2u << 8); break; }
      }
      if (st != INF_OK) break;
      if (bfinal) break;
    }
    if (st == INF_OK && opos != isize) st = INF_SIZE_MISMATCH;
    if (ablate & 0xFFu) st = INF_OK;
    if (lane == 0) status[b] = st;
  }
  if (dbg && lane == 0) {
    atomicAdd(&dbg[0], dbg_rounds);
    atomicAdd(&dbg[1], dbg_passes);
    for (int i = 0; i < 5; i++) atomicAdd((unsigned long long*)(dbg + 2) + i, tc[i]);
    atomicAdd(&dbg[12], dbg_matches);
    atomicAdd(&dbg[13], dbg_near);
#ifdef V2_FIXSTAT
    for (int k = 0; k < 4; k++) { atomicAdd(&dbg[14 + k], fs_lanes[k]); atomicAdd(&dbg[18 + k], fs_iters[k]); }
#endif
  }
}

void v2_guard_report() {
#ifdef V2_GUARD
  unsigned int w[8] = {0};
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(w, HIP_SYMBOL(v2_guard_word), sizeof w);
  fprintf(stderr, "[v2 guard] mask=%#x src_idx=%u lit_opos=%u mpos=%u resolve=%u resolve_win=%u\n", w[0], w[1], w[2], w[3], w[4], w[5]);
#endif
}

int v2_resident_wg_per_cu() {
  int n = 0;
  // resident WAVES per CU (= members decoded concurrently per CU)
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bgzf_inflate_v2, WAVE * V2_WAVES_PER_WG, 0) != hipSuccess || n < 1) n = 8;
  return n * V2_WAVES_PER_WG;
}
static uint32_t g_v2_grid = 0;
void launch_bgzf_inflate_v2(const uint8_t* comp, const uint64_t* blk_coff, const uint64_t* blk_uoff, uint8_t* out,
                            uint32_t n_blocks, uint32_t* status, uint32_t* counter, unsigned long long* scratch,
                            uint32_t scratch_stride, uint32_t grid, uint32_t* dbg, hipStream_t st) {
  if (!n_blocks) return;
  // timing-only phase ablation is a compile-time build (make EXTRA=-DV2_ABLATE=n, tools/ab_ablate.sh): 1 = no resolve,
  // 2 = no write pass.  Such a library produces wrong bytes on purpose and must never ship.
#ifdef V2_ABLATE
  const uint32_t ablate = n_blocks > 64 ? (uint32_t)(V2_ABLATE) & 0xFFu : 0u;
#else
  const uint32_t ablate = 0u;
#endif
  const uint32_t dbg_block = 0xFFFFFFFFu;
  (void)g_v2_grid;
  hipMemsetAsync(counter, 0, 4, st);
  uint32_t g = grid < n_blocks ? grid : n_blocks;
  g = (g + V2_WAVES_PER_WG - 1) / V2_WAVES_PER_WG;  // `grid` counts waves; the scratch holds grid + V2_WAVES_PER_WG match lists
  hipLaunchKernelGGL(k_bgzf_inflate_v2, dim3(g), dim3(WAVE * V2_WAVES_PER_WG), 0, st, comp, blk_coff, blk_uoff, out, n_blocks, status, counter,
                     scratch, scratch_stride, dbg, ablate, dbg_block);
#ifdef V2_GUARD
  v2_guard_report();
#endif
}

}  // namespace bioscan
