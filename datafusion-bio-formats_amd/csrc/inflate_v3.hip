// inflate_v3.hip -- K1: BGZF inflate with wave-parallel Huffman decoding (gfx950, wave64).
//
// Replaces noodles-bgzf 0.49.0 `Reader::read_block` + libdeflate `deflate_decompress` (un-vendored dependency of the
// reference; call sites bio-format-bam/src/storage.rs:161-169, 285-295).  Format: SAM spec 4.1 (BGZF member) + RFC 1951.
//
// One BGZF member per wavefront, persistent grid (waves pull members from an atomic counter).  A DEFLATE block body is
// decoded in ROUNDS; a round cuts the next stretch of compressed bits into 64 sub-streams, one per lane:
//   1. COUNT phase.  Lane 0 starts at the exact bit; lane i > 0 starts `ov` bits before its boundary (almost never a
//      symbol start).  DEFLATE streams self-synchronise, so a lane has usually found the true symbol chain by the time
//      it crosses its boundary; it records its first symbol start there and counts output bytes / matches from that
//      point to the first symbol start at or after its limit.  Fix-point: lane i+1 compares lane i's end with its own
//      recorded start and re-decodes only on mismatch (lane 0 is exact, so by induction the chain is exact --
//      speculation only affects speed).  Lanes after the first END-OF-BLOCK are dead.
//      Sub-streams are LONG (up to V3_MAX_SUB_DW dwords, sized from the previous block's length so that a round covers
//      one DEFLATE block) and the pre-roll is long enough (up to 480 bits) that a lane almost never fails to
//      synchronise: 1.1-1.4 count passes per round instead of the 3.1 of short sub-streams (r01's 7-dword rounds),
//      whose 30 % per-lane failure rate made every round pay two to three fix passes for a handful of lanes.
//   2. CHECKPOINTS.  Every V3_CK_STEPS decode steps (a wave-uniform step counter) each lane stores its decoder state
//      (bit position, table state, byte / match counters) to an L2-resident scratch column.  After the fix-point the
//      checkpoints of a lane cut its sub-stream into SEGMENTS of at most V3_CK_STEPS steps whose exact start state and
//      exact output offset are known.
//   3. WRITE phase in mini-rounds: the segments of the round, in output order, are dealt 64 at a time to the lanes.
//      A mini-round therefore decodes <= 64 x V3_CK_STEPS symbols that produce ONE contiguous range of the output,
//      small enough for the LDS window: literals go to the window, LZ77 matches to the wave's match list, the list is
//      resolved in two walks (sources before the window: straight copies from HBM; sources inside it: dependency-
//      ordered batches of 64) and the window is flushed with coalesced 16-byte stores.
// Tables: zlib-style two-level tables with 16-bit entries in LDS; block headers / code lengths are parsed by a uniform
// register-staged bit reader.  Every synchronisation is wave-local.
#include "kernels.h"
#include <stdlib.h>
#include <stdio.h>

namespace bioscan {

#define WAVE 64
#ifndef V3_SUB_DW
#define V3_SUB_DW 64          // longest sub-stream of a round, dwords
#endif
#ifndef V3_CK_STEPS
#define V3_CK_STEPS 24        // decode steps between two checkpoints = longest segment of the write phase
#endif
#ifndef V3_OV_MAX
#define V3_OV_MAX 480         // pre-roll: half a sub-stream, at least V3_OV_MIN, at most this
#endif
#ifndef V3_OV_MIN
#define V3_OV_MIN 96
#endif
#ifndef V3_OV_QUARTERS
#define V3_OV_QUARTERS 2      // pre-roll = this many quarters of a sub-stream (before the clamps)
#endif
#ifndef V3_WIN_BYTES
#define V3_WIN_BYTES 4096
#endif
constexpr int V3_LIT_BITS = 9;                          // zlib's root sizes: ENOUGH_LENS = 852, ENOUGH_DISTS = 592
constexpr int V3_DIST_BITS = 6;
constexpr int V3_MAX_SUB_DW = V3_SUB_DW;
constexpr int V3_WIN = V3_WIN_BYTES;                    // LDS output window of one round (multiple of 16)
constexpr int V3_LIT_SUB = 352;    // 852 - 512 = 340 sub-table entries at most
constexpr int V3_DIST_SUB = 528;   // 592 - 64
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
constexpr uint32_t V3_NULL_BASE = 2u * ((1u << V3_LIT_BITS) + V3_LIT_SUB + (1u << V3_DIST_BITS) + V3_DIST_SUB) + 64u * 4u;  // byte offset from lit_fast
constexpr uint32_t STOP_END = V3_NULL_BASE, STOP_EOB = V3_NULL_BASE + 2u, STOP_BAD = V3_NULL_BASE + 4u;
// "no such code" is a pointer to the STOP_BAD slot: hitting it parks the lane there, the decode loop has no test for it
constexpr uint32_t E_BAD = E_SUB | ((STOP_BAD >> 1) << 4);
// End-of-block is a pointer to the STOP_EOB slot as well.  Following a pointer consumes the index width of the table
// it sits in, not the code's length: the table build records the difference in V3Lds::eob_fix and the pass subtracts
// it from the lane's end position.
constexpr uint32_t E_EOB = E_SUB | ((STOP_EOB >> 1) << 4);
constexpr uint32_t F_EOB = 1, F_BAD = 2;
// Waves of one workgroup decode different members and never exchange data: a workgroup only exists to get past
// the 16-workgroups-per-CU residency cap (K1 is latency-bound, its speed follows the number of resident waves).
// Every synchronisation is therefore wave-local: LDS operations of one wave execute in order, so a compiler +
// counter fence is all a "barrier" has to be.
#ifndef V3_WAVES_PER_WG
#define V3_WAVES_PER_WG 1     // persistent launches
#endif
#ifndef V3_BOUNDED_WPW
#define V3_BOUNDED_WPW 4      // bounded launches (look-ahead inflate): a retiring workgroup frees room for a 256-thread workgroup
#endif
#define V3_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

// Table-build scratch (code lengths, canonical order, precode table) is only live while a block header
// is parsed, the output window only while a round is written and resolved: they share LDS.
struct V3Build {
  uint16_t lit_sorted[288];
  uint16_t dist_sorted[32];
  uint16_t lit_count[16];
  uint16_t dist_count[16];
  uint16_t t_offs[16], t_first[16], t_w[16];
  uint8_t lens[320];
  uint8_t pre_fast[128];
  uint8_t pre_lens[20];
};
struct __attribute__((aligned(16))) V3Lds {
  uint16_t lit_fast[(1 << V3_LIT_BITS) + V3_LIT_SUB];
  uint16_t dist_fast[(1 << V3_DIST_BITS) + V3_DIST_SUB];  // must follow lit_fast: the decode loop indexes both as one array
  uint32_t be_lut[64];  // [0..31] length symbols 257.., [32..63] distance symbols: base value
  uint16_t null_slot[7];  // must follow be_lut: self-pointing entries a stopped lane idles on (see v3_pass)
  uint16_t eob_fix;       // bits a lane over-consumed when it followed the end-of-block pointer (see E_EOB)
  uint32_t bnd_slot, bnd_budget;  // bounded launches: the scratch stride this wave borrowed, members it may still take
  uint32_t rt_lo, rt_hi, rt_base; // retry launches: members of the current group of 64 still to be decoded (a bit each), the group's first member
  uint32_t pre_lo, pre_hi;        // K0's records (address, or 0), parked here for the same reason as blk_final
  uint32_t blk_final;             // BFINAL of the block being decoded (kept here, not in a register: the kernel is at its SGPR limit)
#ifdef V3_PAD_LDS
  uint32_t pad_lds[V3_PAD_LDS / 4];  // occupancy experiment only
#endif
  union {
    uint8_t win[V3_WIN] __attribute__((aligned(16)));
    V3Build b;
  };
};
static_assert(sizeof(V3Build) <= V3_WIN, "output window must be able to hold the table-build scratch");

__device__ __forceinline__ uint32_t uni2(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t bitrev2(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// ---- uniform register-staged bit reader (as v1) ---------------------------------------------------
// (address space 1: see V3_SRC)
typedef const __attribute__((address_space(1))) uint32_t* v3_gsrc_t;
struct UBits {
  v3_gsrc_t base;
  uint32_t cur, nxt, cidx, wpos;
  uint64_t bb;
  int bc;
};
// start reading at bit `bitpos` counted from the 4-byte aligned pointer `base`
__device__ __forceinline__ void ub_init(UBits& s, const uint32_t* base, uint64_t bitpos, int lane) {
  s.base = (v3_gsrc_t)base;
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

#ifdef V3_GUARD
__device__ unsigned int v3_guard_word[8];
#define V3_G(cond, code, val) ((cond) ? (atomicOr(&v3_guard_word[0], 1u << (code)), atomicMax(&v3_guard_word[code], (unsigned)(val)), true) : false)
#else
#define V3_G(cond, code, val) false
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

// Build the two-level decode table of one alphabet: root table of 2^root_bits entries followed by sub-tables for codes
// longer than root_bits (canonical codes that share a root prefix are contiguous in (len, sym) order, so each sub-table is
// sized by the last = longest code of its group).  Returns 1 if the code is over-subscribed, incomplete in a way
// libdeflate rejects, or the sub-table space is exhausted.  Nothing here is serial in the number of symbols: histogram and
// canonical order by ballots, code-space check by a 15-step scalar recurrence on lane registers, sub-tables one long
// code per lane (groups by comparing root prefixes of neighbours, sizes by a suffix scan -- sub-tables are handed out
// from the END of the sub-table space, so a group's place is known from the groups to its right alone), root entries one
// symbol per lane.
__device__ __forceinline__ int v3_build(V3Lds& L, const uint8_t* lens, int n, uint16_t* fast, uint32_t abs_off, int root_bits, int sub_cap,
                        uint16_t* sorted, bool is_dist, int lane) {
  V3_SYNC();
  for (int i = lane; i < (1 << root_bits) + sub_cap; i += WAVE) fast[i] = (uint16_t)E_BAD;  // bit patterns no code maps to
  // 1. histogram of code lengths: 64 symbols per step, one ballot per length value; lane l keeps count[l]
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
  if (lane < 1 || lane > 15) my_cnt = 0;
  // 2. offsets of the length classes in canonical order (lane l: symbols shorter than l), first code of each class and
  //    the code-space check, all on registers
  uint32_t o;
  const uint32_t my_offs = wave_excl_scan_u32(my_cnt, lane, &o);
  uint32_t my_first = 0;
  {
    uint32_t code = 0;
    int left = 1, over = 0;
#pragma unroll
    for (int l = 1; l <= 15; l++) {
      const uint32_t c = __builtin_amdgcn_readlane(my_cnt, l);
      code <<= 1;
      if (lane == l) my_first = code;
      code += c;
      left = (left << 1) - (int)c;
      if (left < 0) over = 1;
    }
    // incomplete codes: libdeflate (the inflater the reference links) accepts only an empty distance code or a code
    // with a single codeword of length 1 (build_decode_table); everything else that leaves code space unused is invalid
    if (left > 0 && !over) {
      const bool empty_ok = o == 0 && is_dist;
      const bool single_ok = o == 1 && __builtin_amdgcn_readlane(my_cnt, 1) == 1;
      if (!empty_ok && !single_ok) over = 1;
    }
    if (over) return 1;
  }
  if (lane < 16) { L.b.t_offs[lane] = (uint16_t)my_offs; L.b.t_first[lane] = (uint16_t)my_first; }
  // 3. canonical order (by length, then symbol): rank of a symbol inside its length class = symbols of the same length
  //    with a smaller index -> per-chunk ballots with a running base per length
  {
    uint32_t run_base = my_offs;  // lane l tracks length l
    for (int c0 = 0; c0 < n; c0 += WAVE) {
      const int sidx = c0 + lane;
      const int l = sidx < n ? (int)lens[sidx] : 0;
      uint32_t slot = 0;
#pragma unroll
      for (int Lk = 1; Lk <= 15; Lk++) {
        const unsigned long long m = __ballot(l == Lk);
        const uint32_t bk = (uint32_t)__builtin_amdgcn_readlane((int)run_base, Lk);
        if (l == Lk) slot = bk + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == Lk) run_base += (uint32_t)__popcll(m);
      }
      if (l) sorted[slot] = (uint16_t)sidx;
    }
  }
  V3_SYNC();
  // 4. sub-tables: codes longer than root_bits are sorted[k0 .. o), walked 64 at a time from the right
  const uint32_t k0 = root_bits < 15 ? __builtin_amdgcn_readlane(my_offs, root_bits + 1) : o;
  if (k0 < o) {
    uint32_t carry_prefix = 0xFFFFFFFFu, carry_sbits = 0, carry_suffix = 0;  // of the chunk to the right
    int over = 0;
    const uint32_t nchunk = (o - k0 + WAVE - 1) / WAVE;
    for (uint32_t ch = nchunk; ch-- > 0;) {
      const uint32_t k = k0 + ch * WAVE + (uint32_t)lane;
      const bool in = k < o;
      uint32_t sym = 0, len = (uint32_t)root_bits + 1, code = 0, prefix = 0xFFFFFFFEu;
      if (in) {
        sym = sorted[k];
        len = lens[sym];
        code = (uint32_t)L.b.t_first[len] + (k - L.b.t_offs[len]);
        prefix = code >> (len - (uint32_t)root_bits);
      }
      const uint32_t nlanes = o - (k0 + ch * WAVE) < (uint32_t)WAVE ? o - (k0 + ch * WAVE) : (uint32_t)WAVE;
      uint32_t np = (uint32_t)__shfl_down((int)prefix, 1, WAVE);
      if ((uint32_t)lane + 1 >= nlanes) np = carry_prefix;
      const bool is_tail = in && np != prefix;
      const uint32_t sub_len = len - (uint32_t)root_bits;  // 1 .. 9
      // inclusive suffix sum of the tails' sub-table sizes (from the right), carried across chunks
      uint32_t suf = is_tail ? 1u << sub_len : 0u;
#pragma unroll
      for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_down((int)suf, d, WAVE);
        if (lane + d < WAVE) suf += v;
      }
      suf += carry_suffix;
      // the group's tail: the first tail at or to the right of this lane (or the carried group of the next chunk)
      const unsigned long long tmask = __ballot(is_tail) >> lane;
      const int tl = tmask ? lane + __builtin_ctzll(tmask) : lane;
      uint32_t sbits = (uint32_t)__shfl((int)sub_len, tl, WAVE);
      if (!tmask) sbits = carry_sbits;
      if (__ballot(in && suf > (uint32_t)sub_cap) != 0ull) { over = 1; break; }
      const uint32_t base = (1u << root_bits) + (uint32_t)sub_cap - suf;  // entry index of the group's sub-table
      if (is_tail) fast[bitrev2(prefix, root_bits)] = (uint16_t)(E_SUB | ((abs_off + base) << 4) | sbits);  // sbits <= 9 (distance codes)
      if (in) {
        const uint32_t r = bitrev2(code, (int)len) >> root_bits;  // bits after the root, LSB-first
        const uint16_t e = (uint16_t)sym_entry((int)sym, (int)sub_len, is_dist);  // the root bits are consumed when the pointer is followed
        if (!is_dist && sym == 256u) L.eob_fix = (uint16_t)(sbits - sub_len);
        for (uint32_t i = r; i < (1u << sbits); i += (1u << sub_len)) fast[base + i] = e;
      }
      carry_prefix = __builtin_amdgcn_readlane(prefix, 0);
      carry_sbits = __builtin_amdgcn_readlane(sbits, 0);
      carry_suffix = __builtin_amdgcn_readlane(suf, 0);
    }
    if (over) return 1;
  }
  // 5. root entries: one symbol per lane, replicated over the unused high index bits
  for (uint32_t k = lane; k < k0; k += WAVE) {
    const int sym = sorted[k];
    const int l = lens[sym];
    const uint32_t c = (uint32_t)L.b.t_first[l] + (k - L.b.t_offs[l]);
    const uint32_t r = bitrev2(c, l);
    const uint16_t e = (uint16_t)sym_entry(sym, l, is_dist);
    if (!is_dist && sym == 256) L.eob_fix = (uint16_t)(root_bits - l);
    for (uint32_t i = r; i < (1u << root_bits); i += (1u << l)) fast[i] = e;
  }
  V3_SYNC();
  return 0;
}

// ---- the decode step ---------------------------------------------------------------------------------
// One table lookup per loop iteration, every lane, no divergent paths.  A lane's bit window is 32 bits starting at `pos`,
// funnel-shifted (v_alignbit) out of two input dwords d0 (dword wp) and d1; `nxt` is dword wp + 2, prefetched.  A symbol
// consumes <= 28 bits, so pos crosses at most one dword per step.  A lane is a small state machine: `tb/mb` describe its
// next lookup (byte offset of the table from lit_fast, index width).  tb == 0 is the literal/length root (a symbol
// boundary), tb == DIST_BASE the distance root of a pending match, a value below the null slots a sub-table, a null slot
// a stopped lane.  A sub-table pointer consumes the root bits and re-targets the next lookup, a length symbol switches the
// lane to the distance table, a stopped lane follows its self-pointer for ever: lanes in different states share the same
// instructions, so a wave never pays for a path only one lane needs, and there is no loop-carried predicate (each costs
// four scalar instructions per step to merge; K1 is bound by VALU + SALU issue).
// The compressed input is read through an address-space-1 pointer: `base32` is made from an integer (alignment of the payload
// pointer), which hides from the compiler that it points to global memory, and a generic pointer is read with FLAT loads.
// A FLAT load counts in lgkmcnt as well as in vmcnt, so the `s_waitcnt lgkmcnt(0)` behind the table lookup of the NEXT
// decode step waited for the bit-window prefetch to come back from L2 / HBM -- the prefetch hid nothing.
#ifdef V3_GUARD
#define V3_SRC(i) (V3_G((i) > ((limit + 64u) >> 5) + 3u, 1, (i)) ? 0u : gsrc[i])
#else
#define V3_SRC(i) gsrc[i]
#endif
// The lane's bit window: dwords wp (d0) and wp + 1 (d1) feed v_alignbit, the dwords behind them are prefetched.
// V3_REFILL8: two more dwords are held and every SECOND crossing loads 8 bytes -- half the L2 requests of the refill (K1's
// speed follows the number of vector-memory requests per member, not its instruction count: profiles/r03).
#ifdef V3_REFILL8
#define V3_WIN_DECL(wp0) uint32_t d0 = V3_SRC(wp0), d1 = V3_SRC((wp0) + 1), n0 = V3_SRC((wp0) + 2), n1 = V3_SRC((wp0) + 3), n2 = V3_SRC((wp0) + 4)
#define V3_WIN_CROSS()                                                                                   \
  do {                                                                                                   \
    asm volatile("v_mov_b32 %0, %1" : "=v"(d0) : "v"(d1));                                              \
    asm volatile("v_mov_b32 %0, %1" : "=v"(d1) : "v"(n0));                                              \
    asm volatile("v_mov_b32 %0, %1" : "=v"(n0) : "v"(n1));                                              \
    asm volatile("v_mov_b32 %0, %1" : "=v"(n1) : "v"(n2));                                              \
    wp++;                                                                                                \
    if (wp & 1u) {                                                                                       \
      n1 = gsrc[wp + 3]; n2 = gsrc[wp + 4];   /* adjacent: one global_load_dwordx2 */                  \
    }                                                                                                    \
  } while (0)
#else
#define V3_WIN_DECL(wp0) uint32_t d0 = V3_SRC(wp0), d1 = V3_SRC((wp0) + 1), nxt = V3_SRC((wp0) + 2)
// explicit moves: left to the register allocator, the fresh load is copied into place right away and the
// wave waits for it here instead of one crossing later
#define V3_WIN_CROSS()                                                                                   \
  do {                                                                                                   \
    asm volatile("v_mov_b32 %0, %1" : "=v"(d0) : "v"(d1));                                              \
    asm volatile("v_mov_b32 %0, %1" : "=v"(d1) : "v"(nxt));                                             \
    wp++;                                                                                                \
    nxt = V3_SRC(wp + 2);                                                                                \
  } while (0)
#endif
constexpr uint32_t V3_DIST_BASE = 2u * ((1u << V3_LIT_BITS) + V3_LIT_SUB);
constexpr uint32_t V3_CK_ROW = 3u * 64u;  // dwords per checkpoint index: pos[64], acc[64], state[64]
static_assert((uint32_t)V3_CK_MAX * V3_CK_ROW == V3_CK_DWORDS, "kernels.h sizes the checkpoint scratch for V3_CK_MAX rows");

// SYNC pass: a speculative lane decodes from `start` (an arbitrary bit, `ov` bits in front of its boundary) only to find the
// symbol chain: it parks at the first symbol start at or after `count_from` and reports it.  Nothing is counted, so the
// step needs neither the base / extra-bit values nor the accumulators of the count pass (36 instead of 40 vector
// instructions), and because every lane then starts its count pass at its own first symbol, the checkpoint rows of all
// lanes are aligned to the same steps.  Garbage contains END-OF-BLOCK and unassigned codes: such a stop says nothing about
// the block, the lane carries on from the literal/length root.
__device__ __forceinline__ uint32_t v3_sync(V3Lds& L, uint32_t start, uint32_t count_from, uint32_t limit, v3_gsrc_t gsrc) {
  uint32_t pos = start;
  const bool run0 = pos < count_from;  // a lane that starts exactly on its boundary (lane 0: the round's first bit) is there already
  uint32_t wp = pos >> 5;
  const uint32_t wp0 = run0 ? wp : 0u;
  V3_WIN_DECL(wp0);
  const uint8_t* __restrict__ T = (const uint8_t*)L.lit_fast;
  uint32_t tb = run0 ? 0u : STOP_END, mb = run0 ? (uint32_t)V3_LIT_BITS : 0u;
  while (__ballot(tb < V3_NULL_BASE) != 0ull) {
#ifdef V3_ASM_MARKERS
    asm volatile("; V3LOOP_BEGIN %0" ::"n"(3));
#endif
    const bool in_lit = tb < V3_DIST_BASE;
    const uint32_t w = __builtin_amdgcn_alignbit(d1, d0, pos & 31u);
    uint32_t e;
    {
      const uint32_t ea = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)T + tb + (__builtin_amdgcn_ubfe(w, 0u, mb) << 1);
      asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(ea));
    }
    __builtin_assume(e < 65536u);
    const uint32_t l = e & 15u;
    const bool ptr = e >= E_SUB;
    const bool is_lit = e < 0x1000u;
    const bool lenlike = __builtin_amdgcn_inverse_ballot_w64(~(__ballot(is_lit) | __ballot(ptr)));
    const bool is_len = lenlike && in_lit;
    const uint32_t ebv = lenlike ? ((e >> 4) & 15u) : 0u;
    pos += ptr ? mb : l + ebv;
    if ((pos >> 5) != wp) V3_WIN_CROSS();
    // next lookup: a completed symbol at / after count_from is what the lane was looking for
    const bool arrived = pos >= count_from;
    uint32_t ntb = is_len ? V3_DIST_BASE : (arrived ? STOP_END : 0u);
    uint32_t nmb = is_len ? (uint32_t)V3_DIST_BITS : (arrived ? 0u : (uint32_t)V3_LIT_BITS);
    if (ptr) { ntb = (e >> 3) & 0xFFEu; nmb = l; }
    if (ntb > STOP_END) { ntb = arrived ? STOP_END : 0u; nmb = arrived ? 0u : (uint32_t)V3_LIT_BITS; }  // bogus END-OF-BLOCK / bad code
    tb = ntb; mb = nmb;
#ifdef V3_ASM_MARKERS
    asm volatile("; V3LOOP_END %0" ::"n"(3));
#endif
  }
  (void)limit;
  return pos;
}

// COUNT pass of this lane's sub-stream [start, limit), `start` being a symbol start: bytes produced [0:19] and matches
// [20:31] in one accumulator.  Every V3_CK_STEPS steps (wave-uniform counter) a lane that is still running stores
// (pos, acc, tb | mb << 12 | pending match length << 16) to checkpoint row c of the wave's scratch; ck_n = number of the
// lane's valid rows (rows 1 .. ck_n).  Returns true (uniform) when a pass needs more than V3_CK_MAX rows: the caller
// restarts the round with short sub-streams.
#ifdef V3_UTIL
__device__ unsigned long long v3_util[4];  // dev diagnostic: loop iterations / lane-steps of the first count pass, of the fix passes
#endif
__device__ __forceinline__ bool v3_count(V3Lds& L, bool active, uint32_t start, uint32_t limit,
                                         v3_gsrc_t gsrc, uint32_t* __restrict__ ck, int lane,
                                         uint32_t& end_out, uint32_t& acc_out, uint32_t& flags, uint32_t& ck_n, int kind = 0) {
  static_assert(offsetof(V3Lds, null_slot) - offsetof(V3Lds, lit_fast) == V3_NULL_BASE, "null slots must sit at V3_NULL_BASE");
  uint32_t pos = start;
  uint32_t acc = 0;
  const bool run0 = active && pos < limit;
  // (a lane that does not run keeps pos, so it never crosses a dword and never loads again: its three reads are parked at 0)
  uint32_t wp = pos >> 5;
  const uint32_t wp0 = run0 ? wp : 0u;
  V3_WIN_DECL(wp0);
  const uint8_t* __restrict__ T = (const uint8_t*)L.lit_fast;  // dist_fast follows lit_fast in LDS
  uint32_t tb = run0 ? 0u : STOP_END, mb = run0 ? (uint32_t)V3_LIT_BITS : 0u, mlen = 0;
  uint32_t cd = V3_CK_STEPS, c = 0;  // wave-uniform: steps to the next checkpoint, checkpoint row
  uint32_t cn = 0;
  bool overflow = false;
#ifdef V3_UTIL
  uint32_t u_it = 0, u_act = 0;
#endif
  while (__ballot(tb < V3_NULL_BASE) != 0ull) {
#ifdef V3_UTIL
    u_it++; u_act += (uint32_t)__popcll(__ballot(tb < V3_NULL_BASE));
#endif
    if (cd == 0) {
      cd = V3_CK_STEPS;
      c++;
      if (c >= (uint32_t)V3_CK_MAX) { overflow = true; break; }
      if (active && tb < V3_NULL_BASE) {
        uint32_t* q = ck + c * V3_CK_ROW + (uint32_t)lane;
        q[0] = pos; q[64] = acc; q[128] = tb | (mb << 12) | ((mlen & 0x1FFu) << 16);
        cn = c;
      }
    }
    cd--;
#ifdef V3_ASM_MARKERS
    asm volatile("; V3LOOP_BEGIN %0" ::"n"(0));
#endif
    const bool in_lit = tb < V3_DIST_BASE;
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
    if ((pos >> 5) != wp) V3_WIN_CROSS();
    if (is_len) mlen = val + (1u << 20);                   // the match count rides in the same accumulator
    const uint32_t produced = is_lit ? 1u : (is_dist ? mlen : 0u);
    acc += produced;
    // next lookup: a completed symbol returns to the literal/length root, or parks if the sub-stream is used up
    const bool at_end = pos >= limit;
    uint32_t ntb = is_len ? V3_DIST_BASE : (at_end ? STOP_END : 0u);
    uint32_t nmb = is_len ? (uint32_t)V3_DIST_BITS : (at_end ? 0u : (uint32_t)V3_LIT_BITS);
    if (ptr) { ntb = (e >> 3) & 0xFFEu; nmb = l; }
    tb = ntb; mb = nmb;
#ifdef V3_ASM_MARKERS
    asm volatile("; V3LOOP_END %0" ::"n"(0));
#endif
  }
#ifdef V3_UTIL
  if (lane == 0) { atomicAdd(&v3_util[2 * kind], (unsigned long long)u_it); atomicAdd(&v3_util[2 * kind + 1], (unsigned long long)u_act); }
#endif
  if (active) {
    end_out = tb == STOP_EOB ? pos - (uint32_t)L.eob_fix : pos;
    acc_out = acc; ck_n = cn;
    flags = tb == STOP_EOB ? F_EOB : (tb == STOP_BAD ? F_BAD : 0u);
  }
  return overflow;
}

// WRITE pass of one SEGMENT: the lane resumes the decoder state (pos, tb, mb, mlen) of a checkpoint (or the exact start of
// a sub-stream: root state) and decodes until pos reaches `stop_any` (the next checkpoint of that sub-stream: reached
// exactly, in whatever state) or, for a sub-stream's last segment, its ordinary end (first symbol start at / after
// `limit`, or END-OF-BLOCK).  MODE 1: literals to HBM; 2: literals to the LDS window (window byte 0 = output byte
// win_base).  Matches are appended to `mlist` at mpos.  Returns F_BAD when a distance reaches before the output's start.
template <int MODE>
__device__ __forceinline__ uint32_t v3_write(V3Lds& L, bool active, uint32_t pos, uint32_t tb0, uint32_t mb0, uint32_t mlen,
                                             uint32_t limit, uint32_t stop_any, uint8_t* out, uint32_t opos,
                                             unsigned long long* mlist, uint32_t mpos, uint32_t win_base,
                                             v3_gsrc_t gsrc) {
  const bool run0 = active && pos < stop_any;
  uint32_t wp = pos >> 5;
  const uint32_t wp0 = run0 ? wp : 0u;
  V3_WIN_DECL(wp0);
  const uint8_t* __restrict__ T = (const uint8_t*)L.lit_fast;
  uint32_t tb = run0 ? tb0 : STOP_END, mb = run0 ? mb0 : 0u;
  while (__ballot(tb < V3_NULL_BASE) != 0ull) {
#ifdef V3_ASM_MARKERS
    asm volatile("; V3LOOP_BEGIN %0" ::"n"(MODE));
#endif
    const bool in_lit = tb < V3_DIST_BASE;
    const uint32_t w = __builtin_amdgcn_alignbit(d1, d0, pos & 31u);
    uint32_t e;
    {
      const uint32_t ea = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint8_t*)T + tb + (__builtin_amdgcn_ubfe(w, 0u, mb) << 1);
      asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(ea));
    }
    __builtin_assume(e < 65536u);
    const uint32_t l = e & 15u;
    const bool ptr = e >= E_SUB;
    const bool is_lit = e < 0x1000u;
    const bool lenlike = __builtin_amdgcn_inverse_ballot_w64(~(__ballot(is_lit) | __ballot(ptr)));
    const bool is_len = lenlike && in_lit;
    const bool is_dist = lenlike && !is_len;
    const uint32_t base = L.be_lut[(e >> 8) & 63u];
    const uint32_t ebv = lenlike ? ((e >> 4) & 15u) : 0u;
    const uint32_t val = base + __builtin_amdgcn_ubfe(w, l, ebv);
    pos += ptr ? mb : l + ebv;
    if ((pos >> 5) != wp) V3_WIN_CROSS();
    bool bad = false;                                      // (an unassigned code is a pointer to STOP_BAD)
    if (MODE == 1) { if (is_lit && !V3_G(opos >= win_base, 2, opos)) out[opos] = (uint8_t)(e >> 4); }
    if (MODE == 2) { if (is_lit) (L.win - win_base)[opos] = (uint8_t)(e >> 4); }  // base pointer folded: one VALU less than an index subtraction
    if (is_len) mlen = val;
    bool okm = is_dist;
    if (okm && val > opos) { bad = true; okm = false; }
    if (okm && !V3_G(mpos >= V3_ML_ENTRIES, 3, mpos)) { uint2 ent; ent.x = opos; ent.y = mlen | (val << 12); ((uint2*)mlist)[mpos] = ent; }  // = opos | mlen << 32 | val << 44
    mpos += okm ? 1u : 0u;
    opos += is_lit ? 1u : (okm ? mlen : 0u);
    const bool at_end = pos >= limit;
    uint32_t ntb = is_len ? V3_DIST_BASE : (at_end ? STOP_END : 0u);
    uint32_t nmb = is_len ? (uint32_t)V3_DIST_BITS : (at_end ? 0u : (uint32_t)V3_LIT_BITS);
    if (ptr) { ntb = (e >> 3) & 0xFFEu; nmb = l; }
    if (pos >= stop_any) { ntb = STOP_END; nmb = 0u; }     // the next checkpoint: the following segment resumes here
    if (bad) { ntb = STOP_BAD; nmb = 0u; }
    tb = ntb; mb = nmb;
#ifdef V3_ASM_MARKERS
    asm volatile("; V3LOOP_END %0" ::"n"(MODE));
#endif
  }
  return tb == STOP_BAD ? F_BAD : 0u;
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
__device__ __forceinline__ void v3_resolve_batch(uint8_t* out, int lane, int nm, uint32_t m_dst, uint32_t m_len, uint32_t m_dist) {
  bool valid = lane < nm;
  if (valid && V3_G(m_dst + m_len > 65536u || m_dist > m_dst || m_len > 258u, 4, m_dst + m_len)) valid = false;
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
// of its source that precedes R (v3_far_copy) and the matches that also read the window are compacted to the front
// of the list, then only those go through the dependency-ordered copy (v3_near_batch), in dense batches of 64.
// v3_far_issue loads the first 16 bytes of the far part (sources before R are final bytes in HBM); v3_far_finish stores them
// into the window and copies what lies beyond 16 bytes.  Split in two so that the caller can have the next batch's loads
// in flight while it stores this batch's (the load latency was the largest single wait of the resolve).
__device__ __forceinline__ uint32_t v3_far_count(uint32_t R, bool valid, uint32_t m_dst, uint32_t m_len, uint32_t m_dist) {
  const uint32_t src_lo = m_dst - m_dist;
  uint32_t n_far = 0;
  if (valid && src_lo < R) { n_far = R - src_lo; if (n_far > m_len) n_far = m_len; }
  return n_far;
}
__device__ __forceinline__ u32x4 v3_far_issue(const uint8_t* out, uint32_t n_far, uint32_t m_dst, uint32_t m_dist) {
  u32x4 v = {0, 0, 0, 0};
  if (n_far) v = ld16(out + (m_dst - m_dist));  // over-read is inside the (padded) buffer
  return v;
}
__device__ __forceinline__ void v3_far_finish(uint8_t* win, const uint8_t* out, uint32_t R, uint32_t n_far, u32x4 v, uint32_t m_dst, uint32_t m_dist) {
  if (n_far) {
    // LDS takes unaligned 4 / 8-byte stores on gfx950: the bytes go out in the widest pieces that fit
    uint8_t* d = win + (m_dst - R);
    const uint8_t* s = out + (m_dst - m_dist);
    if (n_far >= 16) {
      st8(d, (uint64_t)v.x | ((uint64_t)v.y << 32));
      st8(d + 8, (uint64_t)v.z | ((uint64_t)v.w << 32));
      uint32_t k = 16;
      for (; k + 16 <= n_far; k += 16) {
        const u32x4 w = ld16(s + k);
        st8(d + k, (uint64_t)w.x | ((uint64_t)w.y << 32));
        st8(d + k + 8, (uint64_t)w.z | ((uint64_t)w.w << 32));
      }
      if (k < n_far) { v = ld16(s + k); d += k; n_far -= k; } else n_far = 0;
    }
    const uint32_t rem = n_far;
    if (rem) {
      uint32_t o = 0;
      if (rem & 8) { st8(d, (uint64_t)v.x | ((uint64_t)v.y << 32)); o = 8; }
      if (rem & 4) { st4(d + o, o ? v.z : v.x); o += 4; }
      const uint32_t w = o == 0 ? v.x : o == 4 ? v.y : o == 8 ? v.z : v.w;
      if (rem & 2) { st2(d + o, (uint16_t)w); if (rem & 1) d[o + 2] = (uint8_t)(w >> 16); }
      else if (rem & 1) d[o] = (uint8_t)w;
    }
  }
}

// dependency-ordered copy of the in-window part of <= 64 matches (one per lane, sorted by destination)
__device__ __forceinline__ void v3_near_batch(uint8_t* win, uint32_t R, int lane, int nm, uint32_t m_dst, uint32_t m_len, uint32_t m_dist) {
  bool valid = lane < nm;
  if (valid && V3_G(m_dst + m_len > 65536u || m_dist > m_dst || m_len > 258u || m_dst < R || m_dst + m_len - R > (uint32_t)V3_WIN, 5, m_dst + m_len)) valid = false;
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
  uint32_t n_far = 0;  // already copied by v3_far_copy
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

// Resolve of one mini-round: `tot_m` matches of mlist (sorted by destination) over the output range [R, R + tot_out).
// use_win: the range lives in the LDS window (two walks, see v3_far_copy / v3_near_batch) and is flushed to HBM with
// coalesced 16-byte stores afterwards; otherwise literals are already in HBM and the matches are copied there.
__device__ __forceinline__ void v3_resolve(V3Lds& L, uint8_t* out, unsigned long long* mlist, int lane, uint32_t R, uint32_t tot_out,
                                           uint32_t tot_m, bool use_win, bool dbg, uint32_t& dbg_matches, uint32_t& dbg_near) {
  unsigned long long m_next = 0;
  if ((uint32_t)lane < tot_m) m_next = mlist[lane];
  if (use_win) {
    uint32_t n_near = 0;  // matches that also read this round's window, compacted to the front of the list
    // software pipeline: batch k + 1's list entry and far source are loaded before batch k's bytes are stored
    unsigned long long m = m_next;
    if (WAVE + (uint32_t)lane < tot_m) m_next = mlist[WAVE + lane];
    uint32_t md = (uint32_t)(m & 0xFFFFFFFFull), ml = (uint32_t)((m >> 32) & 0xFFFu), mdist = (uint32_t)(m >> 44);
    uint32_t nf = v3_far_count(R, (uint32_t)lane < tot_m, md, ml, mdist);
    u32x4 fv = v3_far_issue(out, nf, md, mdist);
    for (uint32_t k = 0; k < tot_m; k += WAVE) {
      const uint32_t nmb = tot_m - k < WAVE ? tot_m - k : WAVE;
      const bool valid_m = (uint32_t)lane < nmb;
      // next batch: entry (prefetched one batch earlier), far source load issued now
      const unsigned long long m2 = m_next;
      if (k + 2 * WAVE + (uint32_t)lane < tot_m) m_next = mlist[k + 2 * WAVE + lane];
      const uint32_t md2 = (uint32_t)(m2 & 0xFFFFFFFFull), ml2 = (uint32_t)((m2 >> 32) & 0xFFFu), mdist2 = (uint32_t)(m2 >> 44);
      const uint32_t nf2 = v3_far_count(R, k + WAVE + (uint32_t)lane < tot_m, md2, ml2, mdist2);
      const u32x4 fv2 = v3_far_issue(out, nf2, md2, mdist2);
      // this batch: store
      v3_far_finish(L.win, out, R, nf, fv, md, mdist);
      const bool near = valid_m && md - mdist + ml > R;
      const unsigned long long nmask = __ballot(near);
      if (near) mlist[n_near + __builtin_amdgcn_mbcnt_hi((uint32_t)(nmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nmask, 0u))] = m;  // < k + 64: never a slot still to be read
      n_near += (uint32_t)__popcll(nmask);
      if (dbg) { dbg_matches += nmb; dbg_near += (uint32_t)__popcll(nmask); }
      m = m2; md = md2; ml = ml2; mdist = mdist2; nf = nf2; fv = fv2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if ((uint32_t)lane < n_near) m_next = mlist[lane];
    for (uint32_t k = 0; k < n_near; k += WAVE) {
      const uint32_t nmb = n_near - k < WAVE ? n_near - k : WAVE;
      const unsigned long long m = m_next;
      if (k + WAVE + (uint32_t)lane < n_near) m_next = mlist[k + WAVE + lane];
      const uint32_t md = (uint32_t)(m & 0xFFFFFFFFull), ml = (uint32_t)((m >> 32) & 0xFFFu), mdist = (uint32_t)(m >> 44);
      v3_near_batch(L.win, R, lane, (int)nmb, md, ml, mdist);
    }
    // coalesced flush of the window (16 B per lane; the destination may be unaligned)
    uint8_t* dstp = out + R;
    const uint32_t full = tot_out & ~15u;
    for (uint32_t i = (uint32_t)lane * 16; i < full; i += WAVE * 16) {
      const uint32_t* w = (const uint32_t*)(L.win + i);
      u32x4 v;
      v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
      st16(dstp + i, v);
    }
    if ((uint32_t)lane < (tot_out & 15u)) dstp[full + lane] = L.win[full + lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  } else {
    for (uint32_t k = 0; k < tot_m; k += WAVE) {
      const uint32_t nmb = tot_m - k < WAVE ? tot_m - k : WAVE;
      const unsigned long long m = m_next;
      if (k + WAVE + (uint32_t)lane < tot_m) m_next = mlist[k + WAVE + lane];  // prefetch the next batch
      const uint32_t md = (uint32_t)(m & 0xFFFFFFFFull), ml = (uint32_t)((m >> 32) & 0xFFFu), mdist = (uint32_t)(m >> 44);
      v3_resolve_batch(out, lane, (int)nmb, md, ml, mdist);
    }
  }
}

#ifndef V3_WAVES_PER_EU
#define V3_WAVES_PER_EU 5
#endif
// RETRY: the launch behind K1 v4 (inflate_v4.hip) -- only the members v4 marked INF_RETRY (Huffman codes that need more
// sub-table space than v4's LDS pool; this kernel's tables hold zlib's worst case) are decoded, and when counter[1] says
// there is none the grid leaves at once.
template <int WPW, bool BOUNDED, bool RETRY = false>
__global__ __launch_bounds__(WAVE * WPW, V3_WAVES_PER_EU) void k_bgzf_inflate_v3(const uint8_t* __restrict__ comp,
                                                           const uint64_t* __restrict__ blk_coff,
                                                           const uint64_t* __restrict__ blk_uoff, uint8_t* out_all,
                                                           uint32_t n_blocks, uint32_t* __restrict__ status,
                                                           uint32_t* counter, unsigned long long* scratch,
                                                           uint32_t scratch_stride, uint32_t* dbg, uint32_t* slots, uint32_t n_slots,
                                                           uint32_t per_wave, const uint32_t* __restrict__ pre) {
  __shared__ V3Lds L_all[WPW];
  V3Lds& L = L_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  if constexpr (RETRY) {
    if (uni2(counter[1]) == 0u) return;
    if (lane == 0) { L.bnd_budget = blockIdx.x * 64u; L.rt_lo = 0u; L.rt_hi = 0u; L.rt_base = 0u; }
  }
  // Two launch shapes.  PERSISTENT (slots == nullptr): the grid is what the device holds at once, every wave owns scratch
  // stride blockIdx and pulls members from the atomic counter until none is left.  BOUNDED (slots != nullptr): workgroups
  // of WPW waves, every wave pulls at most `per_wave` members and retires, so a workgroup lives a few milliseconds and
  // then frees WPW wave slots + its LDS in one piece -- room in which the 256-thread workgroups of the HBM-bound stages
  // of the previous chunk (other stream, higher priority) fit while a long inflate runs.  (With one-wave workgroups a
  // freed slot is always taken by the next inflate wave: a 4-wave workgroup of another kernel never finds room.)
  // A bounded wave borrows one of n_slots scratch strides for its lifetime: lane 0 claims a free flag by compare-and-swap,
  // starting at a hashed position; n_slots is twice what the device can hold, so a few probes find one.
  // (The kernel runs at the edge of its register budget: the bounded shape keeps its loop state -- members left, the
  // borrowed stride -- in LDS and is a separate instantiation, so the persistent one compiles to what it was.)
  uint32_t slot = blockIdx.x * WPW + (threadIdx.x >> 6);
  if constexpr (BOUNDED) {
    uint32_t h = 0;
    if (lane == 0) {
      h = (slot * 0x9E3779B1u) % n_slots;
      // every resident wave holds at most one stride and there are more strides than resident waves (n_slots >= 1.05 x what the
      // device holds), so a free one turns up within a few probes; the probe count is bounded anyway.  A wave that finds none
      // takes no member and retires: the grid is exactly ceil(members / (waves x per_wave)), so ITS members stay undecoded --
      // their status keeps the 0xFFFFFFFF the host wrote before the launch and the launch is reported as failed ("no wave took
      // the member", bgzf_source.cpp).  Safe, never seen with the slot count above, and only reachable in the opt-in look-ahead.
      uint32_t tries = 0;
      while (atomicCAS(&slots[h], 0u, 1u) != 0u) {
        h = h + 1u == n_slots ? 0u : h + 1u;
        if (++tries > (1u << 22)) { h = 0xFFFFFFFFu; break; }
      }
      L.bnd_slot = h;
      L.bnd_budget = h == 0xFFFFFFFFu ? 0u : per_wave;
    }
    slot = uni2(h);
    if (slot == 0xFFFFFFFFu) slot = 0;  // (no member will be taken: the scratch pointer is never used)
  }
  // per-wave scratch (L2-resident): the match list of a mini-round, then the checkpoint rows of a round
  unsigned long long* mlist = scratch + (size_t)slot * scratch_stride;
  uint32_t* ck = (uint32_t*)(mlist + V3_ML_ENTRIES);
  uint32_t dbg_rounds = 0, dbg_passes = 0, dbg_matches = 0, dbg_near = 0, dbg_minis = 0, dbg_idle = 0, dbg_hbm = 0;
  // expected length of the next DEFLATE block body: the previous block's; for the first member a wave takes, that member's
  // payload (a BGZF member is usually one block, and an overestimate only idles lanes behind the END-OF-BLOCK while an
  // underestimate costs whole rounds -- with the fixed 12 KiB guess the first member of a wave ran 2.4 extra rounds, which is
  // most of what a launch of a few thousand members costs: 5.4 rounds per member instead of 3)
  uint64_t pred_bits = 0;
#ifdef V3_FIXSTAT
  uint32_t fs_lanes[4] = {0, 0, 0, 0}, fs_iters[4] = {0, 0, 0, 0};  // lanes re-decoded by / runs of the 1st, 2nd, 3rd, later fix pass
#endif
  unsigned long long tc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0 = 0;
#define TICK() (t0 = dbg ? clock64() : 0)
#define TOCK(i) do { if (dbg) { unsigned long long t1 = clock64(); tc[i] += t1 - t0; t0 = t1; } } while (0)
  // Compile-time experiment (tools/build_variant.sh): s_setprio around one phase, so that the SIMD's issue arbiter prefers the
  // waves that are in it.  V3_PRIO_DECODE / _WRITE / _RESOLVE = priority (1..3) of that phase, 0 elsewhere.
#ifdef V3_PRIO_DECODE
#define PRIO_DECODE(on) __builtin_amdgcn_s_setprio((on) ? V3_PRIO_DECODE : 0)
#else
#define PRIO_DECODE(on) do {} while (0)
#endif
#ifdef V3_PRIO_WRITE
#define PRIO_WRITE(on) __builtin_amdgcn_s_setprio((on) ? V3_PRIO_WRITE : 0)
#else
#define PRIO_WRITE(on) do {} while (0)
#endif
#ifdef V3_PRIO_RESOLVE
#define PRIO_RESOLVE(on) __builtin_amdgcn_s_setprio((on) ? V3_PRIO_RESOLVE : 0)
#else
#define PRIO_RESOLVE(on) do {} while (0)
#endif

  // base / extra-bit LUT of the length and distance symbols (RFC 1951 3.2.5), once per wave
  {
    uint32_t base, eb;
    if (lane < 32) { len_base_extra((uint32_t)lane, &base, &eb); L.be_lut[lane] = lane > 28 ? 0u : base; }
    else { dist_base_extra((uint32_t)lane - 32u, &base, &eb); L.be_lut[lane] = lane - 32 > 29 ? 0u : base; }
    if (lane < 7) L.null_slot[lane] = (uint16_t)(E_SUB | (((V3_NULL_BASE >> 1) + (uint32_t)lane) << 4));
    if (lane == 0) { L.pre_lo = (uint32_t)(uintptr_t)pre; L.pre_hi = (uint32_t)((uintptr_t)pre >> 32); }
  }
  V3_SYNC();

  for (;;) {
    if constexpr (BOUNDED) {
      V3_SYNC();
      const uint32_t left = uni2(L.bnd_budget);
      if (left == 0u) break;
      V3_SYNC();
      if (lane == 0) L.bnd_budget = left - 1u;
    }
    uint32_t b = 0;
    if constexpr (RETRY) {
      // (no shared counter: 65 536 atomic pulls to find 80 members took a quarter as long as the launch they repair.)  A wave
      // looks at the status array 64 members at a time -- one coalesced load, one ballot -- in its own stride of such groups, so a
      // small grid finds a handful of members in a few loads per wave; the loop state is parked in LDS like the bounded shape's
      V3_SYNC();
      unsigned long long pend = (unsigned long long)uni2(L.rt_lo) | (unsigned long long)uni2(L.rt_hi) << 32;
      uint32_t gbase = uni2(L.rt_base), nxt = uni2(L.bnd_budget);
      while (pend == 0ull && nxt < n_blocks) {
        gbase = nxt;
        nxt += 64u * gridDim.x;
        const uint32_t bi = gbase + (uint32_t)lane;
        pend = __ballot(bi < n_blocks && status[bi] == (uint32_t)INF_RETRY);
      }
      if (pend == 0ull) break;
      b = gbase + (uint32_t)__builtin_ctzll(pend);
      pend &= pend - 1ull;
      V3_SYNC();
      if (lane == 0) { L.rt_lo = (uint32_t)pend; L.rt_hi = (uint32_t)(pend >> 32); L.rt_base = gbase; L.bnd_budget = nxt; }
      V3_SYNC();
    } else {
      if (lane == 0) b = atomicAdd(counter, 1u);
      b = uni2(b);
      if (b >= n_blocks) break;
    }

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
    // a bounded wave lives for a few members: the block length is predicted from the member itself (a BGZF member is usually one block)
    if (BOUNDED || pred_bits == 0) pred_bits = payload_len * 8 < 2048 ? 2048 : payload_len * 8;
    uint64_t P = skew;
    uint32_t opos = 0;
    bool first_block = true;

    for (;;) {  // DEFLATE blocks
      TICK();
      UBits in;
      ub_init(in, base32, P, lane);
      ub_refill(in, lane);
      const uint32_t bfinal = ub_take(in, 1);
      const uint32_t btype = ub_take(in, 2);
      if (lane == 0) L.blk_final = bfinal;
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
      // A member's first block header may have been parsed by K0 (inflate_headers.hip, one member per lane): the bits the
      // header took and the 320 code lengths as nibbles.  A record K0 did not mark usable means "parse it here".
      uint32_t pre_bits = 0;
      const uint32_t* pr = nullptr;
      if (first_block && btype == 2) {
        pr = (const uint32_t*)((uint64_t)uni2(L.pre_lo) | (uint64_t)uni2(L.pre_hi) << 32);
        if (pr) {
          pr += (size_t)b * V3_PRE_DWORDS;
          if (uni2(pr[0]) & 1u) pre_bits = uni2(pr[1]);
        }
      }
      if (pre_bits) {
        if (lane < 40) {
          const uint32_t w = pr[2 + lane];
          uint32_t lo = w & 0xFFFFu, hi = w >> 16;  // 8 nibbles -> 8 bytes
          lo = (lo | (lo << 8)) & 0x00FF00FFu; lo = (lo | (lo << 4)) & 0x0F0F0F0Fu;
          hi = (hi | (hi << 8)) & 0x00FF00FFu; hi = (hi | (hi << 4)) & 0x0F0F0F0Fu;
          ((uint32_t*)L.b.lens)[2 * lane] = lo;
          ((uint32_t*)L.b.lens)[2 * lane + 1] = hi;
        }
        V3_SYNC();
        ub_init(in, base32, skew + pre_bits, lane);
      } else if (btype == 1) {
        for (int i = lane; i < 320; i += WAVE) {
          uint8_t l;
          if (i < 144) l = 8; else if (i < 256) l = 9; else if (i < 280) l = 7; else if (i < 288) l = 8; else l = 5;
          L.b.lens[i] = l;
        }
        V3_SYNC();
      } else {
        ub_refill(in, lane);
        const uint32_t hlit = ub_take(in, 5) + 257;
        const uint32_t hdist = ub_take(in, 5) + 1;
        const uint32_t hclen = ub_take(in, 4) + 4;
        if (hlit > 286 || hdist > 30) { st = INF_BAD_CODE | (1u << 8); break; }
        // Code-length code (RFC 1951 3.2.7) entirely in registers: lane s holds the length of precode symbol s, the
        // 128-entry decode table lives in two registers (entry i in lane i & 63) and is read with v_readlane, the 320 code
        // lengths being decoded are five registers (symbol j in lane j & 63) -- the serial loop below never waits for LDS.
        uint32_t pl = 0;
        {
          // order of the code-length symbols, 5 bits each: 16 17 18 0 8 7 9 6 10 5 11 4 | 12 3 13 2 14 1 15
          const uint64_t ord0 = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
          const uint64_t ord1 = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
          for (uint32_t i = 0; i < hclen; i++) {
            ub_refill(in, lane);
            const uint32_t v = ub_take(in, 3);
            const uint32_t sidx = (uint32_t)((i < 12 ? ord0 >> (5 * i) : ord1 >> (5 * (i - 12))) & 31u);
            if ((uint32_t)lane == sidx) pl = v;
          }
        }
        uint32_t t_lo = 0, t_hi = 0;  // precode decode table: sym << 3 | len, 0 = no such code
        {
          // canonical codes of the 19 symbols: class counts by ballot, first codes by a 7-step scalar recurrence
          uint32_t code = 0, used = 0, nsym = 0, my_code = 0, cnt1 = 0;
#pragma unroll
          for (int l = 1; l <= 7; l++) {
            const unsigned long long m = __ballot(pl == (uint32_t)l);
            const uint32_t c = (uint32_t)__popcll(m);
            code <<= 1;
            if (pl == (uint32_t)l) my_code = code + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            code += c;
            used += c << (7 - l);
            nsym += c;
            if (l == 1) cnt1 = c;
          }
          // code space of the code-length code: over-subscribed is invalid, incomplete only passes as a single 1-bit code
          if (used > 128u || (used < 128u && !(nsym == 1 && cnt1 == 1))) { st = INF_BAD_CODE | (2u << 8); break; }
          const uint32_t my_rev = pl ? bitrev2(my_code, (int)pl) : 0u;
          for (int sy = 0; sy < 19; sy++) {
            const uint32_t ls = __builtin_amdgcn_readlane(pl, sy);
            if (!ls) continue;
            const uint32_t rc = __builtin_amdgcn_readlane(my_rev, sy);
            const uint32_t mask = (1u << ls) - 1u;
            const uint32_t ent = ((uint32_t)sy << 3) | ls;
            if (((uint32_t)lane & mask) == rc) t_lo = ent;
            if ((((uint32_t)lane + 64u) & mask) == rc) t_hi = ent;
          }
        }
        uint32_t lr0 = 0, lr1 = 0, lr2 = 0, lr3 = 0, lr4 = 0;  // code lengths: literal/length j at j, distance j at 288 + j
        {
          // set `cnt` consecutive lengths of the combined sequence, starting at sequence index i0, to val
          auto set_run = [&](uint32_t i0, uint32_t cnt, uint32_t val) {
            // the sequence runs through the literal/length lengths into the distance lengths (stored from 288)
            const uint32_t n1 = i0 < hlit ? (i0 + cnt <= hlit ? cnt : hlit - i0) : 0u;
            const uint32_t j1 = i0, j2 = 288u + (i0 + n1 - hlit), n2 = cnt - n1;
            const uint32_t ln = (uint32_t)lane;
            if (ln - j1 < n1 || ln - j2 < n2) lr0 = val;
            if (ln + 64u - j1 < n1 || ln + 64u - j2 < n2) lr1 = val;
            if (ln + 128u - j1 < n1 || ln + 128u - j2 < n2) lr2 = val;
            if (ln + 192u - j1 < n1 || ln + 192u - j2 < n2) lr3 = val;
            if (ln + 256u - j1 < n1 || ln + 256u - j2 < n2) lr4 = val;
          };
          const uint32_t total = hlit + hdist;
          uint32_t i = 0, prev = 0;
          int bad = 0;
          while (i < total) {
            ub_refill(in, lane);
            const uint32_t idx = (uint32_t)in.bb & 127u;
            const uint32_t e = idx < 64u ? __builtin_amdgcn_readlane(t_lo, idx) : __builtin_amdgcn_readlane(t_hi, idx - 64u);
            const uint32_t l = e & 7u, sym = e >> 3;
            if (l == 0) { bad = 1; break; }
            ub_take(in, l);
            if (sym < 16) {
              set_run(i, 1, sym);
              prev = sym;
              i++;
            } else {
              uint32_t rep, val;
              if (sym == 16) { if (i == 0) { bad = 1; break; } rep = 3 + ub_take(in, 2); val = prev; }
              else if (sym == 17) { rep = 3 + ub_take(in, 3); val = 0; }
              else { rep = 11 + ub_take(in, 7); val = 0; }
              if (i + rep > total) { bad = 1; break; }
              set_run(i, rep, val);
              if (sym != 16) prev = 0;
              i += rep;
            }
          }
          if (bad) { st = INF_BAD_CODE | (2u << 8); break; }
          L.b.lens[lane] = (uint8_t)lr0; L.b.lens[lane + 64] = (uint8_t)lr1; L.b.lens[lane + 128] = (uint8_t)lr2;
          L.b.lens[lane + 192] = (uint8_t)lr3; L.b.lens[lane + 256] = (uint8_t)lr4;
          V3_SYNC();
        }
      }
      TOCK(5);
      if (v3_build(L, L.b.lens, 288, L.lit_fast, 0u, V3_LIT_BITS, V3_LIT_SUB, L.b.lit_sorted, false, lane)) { st = INF_BAD_CODE | (3u << 8); break; }
      if (v3_build(L, L.b.lens + 288, 32, L.dist_fast, (1u << V3_LIT_BITS) + V3_LIT_SUB, V3_DIST_BITS, V3_DIST_SUB, L.b.dist_sorted, true, lane)) { st = INF_BAD_CODE | (4u << 8); break; }
      P = ub_bitpos(in);
      TOCK(0);

      // ---- rounds over the block body ----
      bool block_done = false;
      uint32_t force_dw = 0;  // != 0: the round is being re-run with short sub-streams (a pass ran out of checkpoint rows)
      const uint64_t block_P0 = P;
      while (!block_done) {
        // A round should end with its block: it covers what is left of the predicted block length (the previous block's
        // length -- zlib and libdeflate cut blocks of similar size), with a little slack because an underestimate costs a
        // whole extra round and an overestimate only idle lanes behind the END-OF-BLOCK.
        const uint64_t rem_bits = end_bits > P ? end_bits - P : 0;
        const uint64_t used = P - block_P0;
        uint64_t want = pred_bits > used + pred_bits / 8 ? pred_bits - used : pred_bits / 8;
        want += want / 16 + 64;
        if (want > rem_bits) want = rem_bits;
#ifdef V3_EXACT_FINAL
        // Compile-time experiment, measured and NOT the default: a member's FINAL block ends with its payload (at most 7 bits of
        // padding behind the END-OF-BLOCK), so its length needs no prediction -- the rounds that are left share the remaining
        // bits evenly and the last one ends with the block.  Config 2: 2.07 rounds and 2.8 decode passes per member instead
        // of 2.97 and 5.9 (the predictor's 6 % of slack does not cover the variation between members, so most members run a
        // third, short round with a fix pass), 1.1 instead of 3.0 lanes idle behind the END-OF-BLOCK -- and 2 % SLOWER
        // (54.1 against 52.8 ms on 262 144 members): the passes it removes are the cheap ones (short sub-streams), while two
        // equal rounds of 50 dwords balance their lanes a little worse than one of 64 and one of 42.
        if (uni2(L.blk_final)) {
          const uint32_t round_max = 64u * 32u * (uint32_t)V3_MAX_SUB_DW;   // (a payload is < 2^19 bits)
          const uint32_t rb = (uint32_t)rem_bits;
          const uint32_t rounds_left = (rb + round_max - 1u) / round_max;
          want = rounds_left > 1u ? (rb + rounds_left - 1u) / rounds_left : rb;
        }
#endif
        uint32_t sub_dw = (uint32_t)((want + 64ull * 32 - 1) / (64ull * 32));
        if (sub_dw > (uint32_t)V3_MAX_SUB_DW) sub_dw = V3_MAX_SUB_DW;
        if (sub_dw < 5) sub_dw = 5;
        if (force_dw) sub_dw = force_dw;
        const uint32_t subb = sub_dw * 32;
        uint32_t ovb = (subb * (uint32_t)V3_OV_QUARTERS) >> 2;   // pre-roll of the speculative lanes
        if (ovb < (uint32_t)V3_OV_MIN) ovb = V3_OV_MIN;
        if (ovb > (uint32_t)V3_OV_MAX) ovb = V3_OV_MAX;
        const uint64_t wb = P >> 5;
        v3_gsrc_t gsrc = (v3_gsrc_t)(base32 + wb);
        TOCK(1);
        PRIO_DECODE(1);
        const uint32_t rel0 = (uint32_t)(P & 31);
        const uint32_t bnd = rel0 + (uint32_t)lane * subb;
        const uint32_t limit = rel0 + (uint32_t)(lane + 1) * subb;
        uint32_t start = bnd, end = bnd, acc = 0, flags = 0, cn = 0;
        bool ovf;
        {
          // lanes too close to the round's start for a full pre-roll begin at the round's exact first bit instead
          const uint32_t room = (uint32_t)lane * subb;
          const uint32_t ov = room < ovb ? room : ovb;
          start = v3_sync(L, bnd - ov, bnd, limit, gsrc);   // first symbol start at / after the lane's boundary (speculative)
          TOCK(6);
          if (start >= limit) { end = start; }               // (a symbol that spans the whole sub-stream: the lane owns nothing)
          ovf = v3_count(L, start < limit, start, limit, gsrc, ck, lane, end, acc, flags, cn);
        }
        dbg_passes++;
        for (int it = 0; it < 66 && !ovf; it++) {
          const unsigned long long stopm = __ballot(flags != 0);
          const int first_stop = stopm ? __builtin_ctzll(stopm) : 64;
          uint32_t pe = __shfl_up(end, 1, WAVE);
          const bool alive = lane > 0 && lane <= first_stop;
          const bool changed = alive && pe != start;
          if (__ballot(changed) == 0ull) break;
#ifdef V3_FIXSTAT
          { const int k = it < 3 ? it : 3; fs_lanes[k] += (uint32_t)__popcll(__ballot(changed)); fs_iters[k]++; }
#endif
          if (changed) start = pe;
          // a lane whose corrected start already lies beyond its limit owns no symbols
          if (changed && start >= limit) { end = start; acc = 0; flags = 0; cn = 0; }
          ovf = v3_count(L, changed && start < limit, start, limit, gsrc, ck, lane, end, acc, flags, cn, 1);
          dbg_passes++;
        }
        if (ovf) {
          // only reachable with codes of ~2 bits per symbol over a long sub-stream: 7-dword sub-streams always fit
          if (force_dw) { st = INF_OVERRUN | (3u << 8); break; }
          force_dw = 7;
          continue;
        }
        force_dw = 0;
        PRIO_DECODE(0);
        TOCK(2);
        const unsigned long long stopm = __ballot(flags != 0);
        const int last = stopm ? __builtin_ctzll(stopm) : 63;
        const uint32_t last_flags = __builtin_amdgcn_readlane(flags, last);
        if (stopm && (last_flags & F_BAD)) { st = INF_BAD_CODE | (5u << 8); break; }
        const bool valid = lane <= last;
        uint32_t tot_out, tot_m;
        const uint32_t obase = opos + wave_excl_scan_u32(valid ? (acc & 0xFFFFFu) : 0u, lane, &tot_out);
        const uint32_t mbase = wave_excl_scan_u32(valid ? (acc >> 20) : 0u, lane, &tot_m);
        if (opos + tot_out > isize) { st = INF_OVERRUN; break; }
        if (dbg) dbg_idle += 63u - (uint32_t)last;
        // ---- write phase: the round's segments, in output order, 64 per mini-round ----
        const uint32_t nseg = (valid && start < limit) ? 1u + cn : 0u;
        uint32_t n_seg_tot;
        const uint32_t segbase = wave_excl_scan_u32(nseg, lane, &n_seg_tot);
        const uint32_t segend = segbase + nseg;
        uint32_t n_take = 0;  // segments of the current mini-round: 64, or as many as fit in the LDS window
        for (uint32_t s0 = 0; s0 < n_seg_tot; s0 += n_take) {
          const uint32_t g = s0 + (uint32_t)lane;
          bool has = g < n_seg_tot;
          // owner of segment g: the first lane whose segments end after g (segend is non-decreasing)
          int lo = 0, hi = WAVE;
#pragma unroll
          for (int step = 0; step < 7; step++) {  // 65 possible answers
            const int mid = (lo + hi) >> 1;
            const uint32_t v = (uint32_t)__shfl((int)segend, mid & 63, WAVE);
            if (lo < hi) { if (v <= g) lo = mid + 1; else hi = mid; }
          }
          const int own = lo < WAVE ? lo : WAVE - 1;
          const uint32_t k = g - (uint32_t)__shfl((int)segbase, own, WAVE);
          const uint32_t o_start = (uint32_t)__shfl((int)start, own, WAVE);
          const uint32_t o_obase = (uint32_t)__shfl((int)obase, own, WAVE);
          const uint32_t o_mbase = (uint32_t)__shfl((int)mbase, own, WAVE);
          const uint32_t o_cn = (uint32_t)__shfl((int)cn, own, WAVE);
          const uint32_t o_acc = (uint32_t)__shfl((int)acc, own, WAVE);
          const uint32_t o_limit = rel0 + (uint32_t)(own + 1) * subb;
          uint32_t p0 = o_start, a0 = 0, st0 = (uint32_t)V3_LIT_BITS << 12, p1 = 0xFFFFFFFFu, a1 = o_acc;
          if (has && k > 0) {
            const uint32_t* q = ck + k * V3_CK_ROW + (uint32_t)own;        // row k = state after k * V3_CK_STEPS steps
            p0 = q[0]; a0 = q[64]; st0 = q[128];
          }
          if (has && k < o_cn) {
            const uint32_t* q = ck + (k + 1u) * V3_CK_ROW + (uint32_t)own;
            p1 = q[0]; a1 = q[64];
          }
          const uint32_t seg_out = has ? (a1 & 0xFFFFFu) - (a0 & 0xFFFFFu) : 0u;
          const uint32_t seg_m = has ? (a1 >> 20) - (a0 >> 20) : 0u;
          const uint32_t my_opos = o_obase + (a0 & 0xFFFFFu);
          const uint32_t my_mabs = o_mbase + (a0 >> 20);
          const uint32_t R = __builtin_amdgcn_readlane(my_opos, 0);  // first output byte of the mini-round
          const uint32_t M0 = __builtin_amdgcn_readlane(my_mabs, 0);
          // Segments are consecutive in the output, so those whose bytes end inside the LDS window are a prefix of the
          // lanes: the mini-round takes that prefix (normally all 64) and the next one starts behind it.  A single segment
          // larger than the window (forty 258-byte matches) goes through HBM on its own.
          const uint32_t n_fit = (uint32_t)__popcll(__ballot(has && my_opos + seg_out - R <= (uint32_t)V3_WIN));
          const bool use_win = n_fit != 0u;
          n_take = use_win ? n_fit : 1u;
          has = has && (uint32_t)lane < n_take;
          const uint32_t nl = n_take - 1u;  // last lane with a segment
          const uint32_t out_s = __builtin_amdgcn_readlane(my_opos + seg_out, nl) - R;
          const uint32_t m_s = __builtin_amdgcn_readlane(my_mabs + seg_m, nl) - M0;
          if (m_s > (uint32_t)V3_ML_ENTRIES) { st = INF_OVERRUN | (1u << 8); break; }
          if (dbg && !use_win) dbg_hbm++;
          {
            uint32_t f2 = 0;
            const uint32_t tb0 = st0 & 0xFFFu, mb0 = (st0 >> 12) & 15u, ml0 = st0 >> 16;
#ifndef V3_ABLATE_WRITE
            PRIO_WRITE(1);
            if (use_win) f2 = v3_write<2>(L, has, p0, tb0, mb0, ml0, o_limit, p1, out, my_opos, mlist, my_mabs - M0, R, gsrc);
#ifdef V3_GUARD
            else f2 = v3_write<1>(L, has, p0, tb0, mb0, ml0, o_limit, p1, out, my_opos, mlist, my_mabs - M0, isize, gsrc);
#else
            else f2 = v3_write<1>(L, has, p0, tb0, mb0, ml0, o_limit, p1, out, my_opos, mlist, my_mabs - M0, R, gsrc);
#endif
#endif
            PRIO_WRITE(0);
            dbg_minis++;
            if (__ballot(f2 & F_BAD) != 0ull) { st = INF_BAD_DIST; break; }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          TOCK(3);
#ifndef V3_ABLATE_RESOLVE
          PRIO_RESOLVE(1);
          v3_resolve(L, out, mlist, lane, R, out_s, m_s, use_win, dbg != nullptr, dbg_matches, dbg_near);
          PRIO_RESOLVE(0);
#endif
          TOCK(4);
        }
        if (st != INF_OK) break;
        opos += tot_out;
        dbg_rounds++;
        const uint32_t end_last = __builtin_amdgcn_readlane(end, last);
        P = (wb << 5) + end_last;
        if (stopm) block_done = true;
        else if (P >= end_bits + 64) { st = INF_OVERRUN | (2u << 8); break; }
      }
      if (st == INF_OK) {
        // the next block is expected to be as long as this one; a member's short last block is not a good predictor
        // unless it is the member's only block
        const uint64_t blk_bits = P - block_P0;
        if (!bfinal || first_block) pred_bits = blk_bits < 2048 ? 2048 : blk_bits;
      }
      first_block = false;
      if (st != INF_OK) break;
      if (bfinal) break;
    }
    if (st == INF_OK && opos != isize) st = INF_SIZE_MISMATCH;
    // every block header and every symbol must lie inside the member's payload (libdeflate: reading past the input is bad
    // data): a member whose last block lost its BFINAL bit would otherwise go on into the trailer, which may parse as one
    // more, empty, final block
    if (st == INF_OK && P > end_bits) st = INF_OVERRUN | (4u << 8);
#if defined(V3_ABLATE_WRITE) || defined(V3_ABLATE_RESOLVE)
    st = INF_OK;  // timing-only build: the bytes are wrong on purpose
#endif
    if (lane == 0) status[b] = st;
  }
  if constexpr (BOUNDED) {
    V3_SYNC();
    if (lane == 0 && L.bnd_slot != 0xFFFFFFFFu) atomicExch(&slots[L.bnd_slot], 0u);  // (the scratch carries nothing from one owner to the next)
  }
  if (dbg && lane == 0) {
    atomicAdd(&dbg[0], dbg_rounds);
    atomicAdd(&dbg[1], dbg_passes);
    for (int i = 0; i < 5; i++) atomicAdd((unsigned long long*)(dbg + 2) + i, tc[i]);
    atomicAdd((unsigned long long*)(dbg + 26), tc[5]);
    atomicAdd((unsigned long long*)(dbg + 28), tc[6]);
    atomicAdd(&dbg[12], dbg_matches);
    atomicAdd(&dbg[13], dbg_near);
    atomicAdd(&dbg[22], dbg_minis);
    atomicAdd(&dbg[23], dbg_idle);
    atomicAdd(&dbg[24], dbg_hbm);
#ifdef V3_FIXSTAT
    for (int k = 0; k < 4; k++) { atomicAdd(&dbg[14 + k], fs_lanes[k]); atomicAdd(&dbg[18 + k], fs_iters[k]); }
#endif
  }
}

void v3_guard_report() {
#ifdef V3_GUARD
  unsigned int w[8] = {0};
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(w, HIP_SYMBOL(v3_guard_word), sizeof w);
  fprintf(stderr, "[v2 guard] mask=%#x src_idx=%u lit_opos=%u mpos=%u resolve=%u resolve_win=%u\n", w[0], w[1], w[2], w[3], w[4], w[5]);
#endif
}

int v3_resident_wg_per_cu() {
  int n = 0;
  // resident WAVES per CU (= members decoded concurrently per CU)
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bgzf_inflate_v3<V3_WAVES_PER_WG, false>, WAVE * V3_WAVES_PER_WG, 0) != hipSuccess || n < 1) n = 8;
  return n * V3_WAVES_PER_WG;
}
void launch_bgzf_inflate_v3(const uint8_t* comp, const uint64_t* blk_coff, const uint64_t* blk_uoff, uint8_t* out,
                            uint32_t n_blocks, uint32_t* status, uint32_t* counter, unsigned long long* scratch,
                            uint32_t grid, uint32_t* dbg, hipStream_t st, uint32_t* slots, uint32_t n_slots, uint32_t per_wave, uint32_t wpw,
                            const uint32_t* pre, bool retry_only) {
  if (!n_blocks) return;
  if (retry_only) {
    // (persistent shape only; counter[0] is this launch's member counter, counter[1] the retry count v4 left)
    (void)hipMemsetAsync(counter, 0, 4, st);
    uint32_t g = grid < n_blocks ? grid : n_blocks;
    hipLaunchKernelGGL((k_bgzf_inflate_v3<1, false, true>), dim3(g), dim3(WAVE), 0, st, comp, blk_coff, blk_uoff, out, n_blocks,
                       status, counter, scratch, (uint32_t)V3_SCRATCH_STRIDE, nullptr, nullptr, 0u, 0u, nullptr);
    return;
  }
  if (wpw != 1) wpw = V3_BOUNDED_WPW;
  (void)hipMemsetAsync(counter, 0, 4, st);
  if (slots) {
    // bounded: workgroups of `wpw` waves, `per_wave` members each; scratch strides handed out through `slots`
    if (!per_wave) per_wave = 1;
    const uint32_t per_wg = wpw * per_wave;
    const uint32_t g = (n_blocks + per_wg - 1) / per_wg;
    if (wpw == 1)
      hipLaunchKernelGGL((k_bgzf_inflate_v3<1, true>), dim3(g), dim3(WAVE), 0, st, comp, blk_coff, blk_uoff, out, n_blocks,
                         status, counter, scratch, (uint32_t)V3_SCRATCH_STRIDE, dbg, slots, n_slots, per_wave, pre);
    else
      hipLaunchKernelGGL((k_bgzf_inflate_v3<V3_BOUNDED_WPW, true>), dim3(g), dim3(WAVE * V3_BOUNDED_WPW), 0, st, comp, blk_coff, blk_uoff, out, n_blocks,
                         status, counter, scratch, (uint32_t)V3_SCRATCH_STRIDE, dbg, slots, n_slots, per_wave, pre);
  } else {
    uint32_t g = grid < n_blocks ? grid : n_blocks;
    g = (g + V3_WAVES_PER_WG - 1) / V3_WAVES_PER_WG;  // `grid` counts waves; the scratch holds grid + V3_WAVES_PER_WG strides
    hipLaunchKernelGGL((k_bgzf_inflate_v3<V3_WAVES_PER_WG, false>), dim3(g), dim3(WAVE * V3_WAVES_PER_WG), 0, st, comp, blk_coff, blk_uoff, out, n_blocks,
                       status, counter, scratch, (uint32_t)V3_SCRATCH_STRIDE, dbg, nullptr, 0u, 0u, pre);
  }
#ifdef V3_GUARD
  v3_guard_report();
#endif
#ifdef V3_UTIL
  {
    unsigned long long u[4] = {0, 0, 0, 0};
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(u, HIP_SYMBOL(v3_util), sizeof u);
    if (n_blocks > 1000)
      fprintf(stderr, "[v3 util] first count pass: %llu steps, %.1f lanes busy per step; fix passes: %llu steps, %.1f lanes busy per step (cumulative)\n",
              u[0], u[0] ? (double)u[1] / u[0] : 0.0, u[2], u[2] ? (double)u[3] / u[2] : 0.0);
  }
#endif
}

}  // namespace bioscan
