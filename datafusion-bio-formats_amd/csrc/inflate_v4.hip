// inflate_v4.hip -- K1: BGZF inflate with wave-parallel Huffman decoding (gfx950, wave64).
//
// Replaces noodles-bgzf 0.49.0 `Reader::read_block` + libdeflate `deflate_decompress` (un-vendored dependency of the
// reference; call sites bio-format-bam/src/storage.rs:161-169, 285-295).  Format: SAM spec 4.1 (BGZF member) + RFC 1951.
//
// One BGZF member per wavefront, persistent grid (waves pull members from an atomic counter).  A DEFLATE block body is
// decoded in ROUNDS; a round cuts the next stretch of compressed bits into 64 long sub-streams, one per lane:
//   1. SYNC pass: a lane starts up to V4_OV_MAX bits before its boundary and decodes only to find the symbol chain.
//   2. COUNT pass with fix-point: every lane decodes its sub-stream from its first symbol, counting output bytes and
//      matches; lane i+1 re-decodes only when lane i's end differs from its own start (lane 0 is exact, so by induction
//      the chain is exact).  Every V4_CK_STEPS steps a lane stores its state to a checkpoint row.
//   3. WRITE phase in mini-rounds: the segments between checkpoints, in output order, 64 at a time: literals to the LDS
//      window, matches to the match list, LZ77 resolve in two walks, coalesced flush.
// What is new against v3 (r02/r03) is the DECODE STEP.  v3's 16-bit table entries had to be classified and the next
// lookup derived from the class on the vector ALU (33 / 53 / 53 vector instructions per step in the sync / count / write
// loops, K1 being bound by vector-instruction issue); v4's 32-bit entries DESCRIBE THE NEXT LOOKUP THEMSELVES -- the
// table it goes to, its index width and the bits this step consumes are fields of the entry, the length / distance value
// is (m << eb) + extra bits with m and eb in the entry (no second, dependent LDS read of a base-value table) -- and a
// lane that is done leaves the loop through EXEC instead of idling on a null slot.
#include "kernels.h"
#include <stdlib.h>
#include <stdio.h>

namespace bioscan {

#define WAVE 64
#ifndef V4_SUB_DW
#define V4_SUB_DW 11          // longest sub-stream of a round, dwords: a round's compressed bits are staged in LDS (64 x this)
#endif
#ifndef V4_CK_STEPS
#define V4_CK_STEPS 20        // decode steps between two checkpoints = longest segment of the write phase
#endif
#ifndef V4_OV_MAX
#define V4_OV_MAX 480         // pre-roll: half a sub-stream, at least V4_OV_MIN, at most this
#endif
#ifndef V4_OV_MIN
#define V4_OV_MIN 96
#endif
#ifndef V4_OV_QUARTERS
#define V4_OV_QUARTERS 8      // pre-roll = this many quarters of a sub-stream (before the clamps)
#endif
#ifndef V4_WIN_BYTES
#define V4_WIN_BYTES 2496
#endif
#ifndef V4_NO_ASM
#define V4_NO_ASM 0           // 1: the C++ forms of the sync / count loops everywhere (what the bounded launch shape always runs)
#endif
#define V4_CK_MAX V3_CK_MAX   // the scratch layout (match list + checkpoint rows) is v3's: kernels.h
constexpr uint32_t V4_ML_ENTRIES = V3_ML_ENTRIES, V4_CK_DWORDS = V3_CK_DWORDS, V4_SCRATCH_STRIDE = V3_SCRATCH_STRIDE, V4_PRE_DWORDS = V3_PRE_DWORDS;
constexpr int V4_LIT_BITS = 9;                          // zlib's root sizes: ENOUGH_LENS = 852, ENOUGH_DISTS = 592
constexpr int V4_DIST_BITS = 6;
constexpr int V4_MAX_SUB_DW = V4_SUB_DW;
constexpr int V4_WIN = V4_WIN_BYTES;                    // LDS output window of one round (multiple of 16)
// The compressed bits of a round live in LDS while it is decoded (three passes over them): 64 sub-streams plus the dwords a
// lane may look at behind the last sub-stream's end (a symbol that begins before the limit, the two prefetched dwords).
#ifndef V4_LCAP_N
#define V4_LCAP_N 256
#endif
constexpr uint32_t V4_LCAP = V4_LCAP_N;                      // matches of a mini-round (together with the window size: what a mini-round takes)
#ifndef V4_PIPE_CK
#define V4_PIPE_CK 0    // 1: a mini-round's checkpoint rows are asked for one mini-round ahead (measured: no gain -- see below)
#endif
#ifndef V4_PREFETCH
#define V4_PREFETCH 1   // 0: every round waits for its own staging loads (A/B switch)
#endif
constexpr uint32_t V4_STAGE_SLACK = 16, V4_STAGE_DW = 64u * V4_SUB_DW + V4_STAGE_SLACK;
// One table of 1076 32-bit entries, addressed in entry units (all table starts are even):
//   [0, 64)             distance root
//   [64, 560)           the sub-table POOL: distance sub-tables from its low end up, literal/length sub-tables from its
//                       high end down (zlib's worst cases are 528 and 340 entries; 20 000 members of config 2 need 262 in the
//                       median, 360 at the 99th percentile and 492 at most, htslib-written BAM 40 + 186 --
//                       tools/experiments/subtable_need.py; the pool is what 16 waves per CU leave: 10 240 B of LDS each).  A block whose codes do not fit is not decoded
//                       here: the member is marked INF_RETRY and the wide-table kernel (inflate_v3.hip) takes it.
//   [560, 1072)         literal/length root
//   [1072, 1074)        STOP_EOB                       [1074, 1076)  STOP_BAD
// Entry:  nmb [3:0]   index width of the NEXT lookup
//         ntb [14:4]  entry offset of the table the NEXT lookup goes to (even, so bit 4 is 0 and v_bfe_u32 can take the
//                     entry itself as its width operand); V4_LIT_ROOT = "a symbol is complete", 0 = the distance root =
//                     "that was a length symbol"
//         adv [19:15] bits this step consumes (code bits + extra bits; a sub-table pointer: the index width of the table
//                     it sits in)
//         eb  [23:20] extra-bit count of a length / distance symbol
//         m   [31:24] literal byte | (length base - 3) >> eb | (distance base - 1) >> eb   (RFC 1951 3.2.5: every base is
//                     m << eb plus 3 resp. 1, with m < 256)
// Which alphabet a table belongs to is its place: distance tables lie below the lowest literal/length sub-table (`lit_lo`,
// per block).  END-OF-BLOCK points to STOP_EOB, a bit pattern no code maps to and the symbols that must not occur (286, 287,
// distance 30, 31) to STOP_BAD.  The count and write loops retire a lane that is sent to a STOP table; the sync pass, which
// decodes garbage on purpose, follows it: the STOP entries lead back to the root without consuming anything (BAD itself
// consumes one bit, so a lane always moves on).
constexpr uint32_t V4_DIST_ROOT = 0, V4_POOL_LO = 1u << V4_DIST_BITS, V4_POOL = 496, V4_LIT_ROOT = V4_POOL_LO + V4_POOL,
                   V4_STOP_EOB = V4_LIT_ROOT + (1u << V4_LIT_BITS), V4_STOP_BAD = V4_STOP_EOB + 2u, V4_NENT = V4_STOP_BAD + 2u;
static_assert((V4_LIT_ROOT & 1u) == 0 && (V4_STOP_EOB & 1u) == 0 && V4_NENT < 2048u, "table starts are even and fit the 11-bit field");
__host__ __device__ constexpr uint32_t v4_enc(uint32_t nmb, uint32_t ntb, uint32_t adv, uint32_t eb, uint32_t m) {
  return nmb | (ntb << 4) | (adv << 15) | (eb << 20) | (m << 24);
}
constexpr uint32_t E4_BAD = v4_enc(0, V4_STOP_BAD, 1, 0, 0);
constexpr uint32_t E4_PASS = v4_enc(V4_LIT_BITS, V4_LIT_ROOT, 0, 0, 0);  // STOP table entries: back to the root (sync pass only)
constexpr uint32_t F_EOB = 1, F_BAD = 2;
// Waves of one workgroup decode different members and never exchange data: a workgroup only exists to get past
// the 16-workgroups-per-CU residency cap.  Every synchronisation is therefore wave-local: LDS operations of one wave
// execute in order, so a compiler + counter fence is all a "barrier" has to be.
#ifndef V4_WAVES_PER_WG
#define V4_WAVES_PER_WG 1     // persistent launches
#endif
#ifndef V4_BOUNDED_WPW
#define V4_BOUNDED_WPW 4      // bounded launches (look-ahead inflate): a retiring workgroup frees room for a 256-thread workgroup
#endif
#define V4_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)

// Table-build scratch (code lengths, canonical order, precode table) is only live while a block header
// is parsed, the output window only while a round is written and resolved: they share LDS.
struct V4Build {
  uint16_t lit_sorted[288];
  uint16_t dist_sorted[32];
  uint16_t lit_count[16];
  uint16_t dist_count[16];
  uint16_t t_offs[16], t_first[16], t_w[16];
  uint8_t lens[320];
  uint8_t pre_fast[128];
  uint8_t pre_lens[20];
};
struct __attribute__((aligned(16))) V4Lds {
  uint32_t tab[V4_NENT];
  uint32_t stage[V4_STAGE_DW] __attribute__((aligned(16)));  // the round's compressed dwords (dword 0 = the dword that holds the round's first bit)
  uint32_t bnd_slot, bnd_budget;  // bounded launches: the scratch stride this wave borrowed, members it may still take
  uint32_t pre_lo, pre_hi;        // K0's records (address, or 0), parked here for the same reason as blk_final
  uint32_t blk_final;             // BFINAL of the block being decoded (kept here, not in a register)
  // The match list of a mini-round whose output lives in the window never leaves the chip: a match's length - 3 and
  // distance - 1 are written into its own first three destination bytes (a match is at least 3 bytes long; 8 + 15 bits), and
  // this list only says where the matches are (window-relative destination, in output order).  Through a list in global
  // memory the resolve waited ~14 us per mini-round for the write pass's stores to drain and the entries to come back
  // (a third of K1's wave cycles), and the list was 8.8 of K1's 33 GB of HBM traffic per 65 536 members.
  uint16_t ml16[V4_LCAP];
  uint32_t lit_lo;                // entry offset of the lowest literal/length sub-table of the block: tables below it are distance tables
#ifdef V4_PAD_LDS
  uint32_t pad_lds[V4_PAD_LDS / 4];  // occupancy experiment only
#endif
  union {
    uint8_t win[V4_WIN + 16] __attribute__((aligned(16)));  // (+ 16: the near copy reads whole 16-byte chunks)
    V4Build b;
  };
};
static_assert(sizeof(V4Build) <= V4_WIN, "output window must be able to hold the table-build scratch");

__device__ __forceinline__ uint32_t uni2(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t bitrev2(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// ---- uniform register-staged bit reader (as v1) ---------------------------------------------------
// (address space 1: see V4_SRC)
typedef const __attribute__((address_space(1))) uint32_t* v4_gsrc_t;
struct UBits {
  v4_gsrc_t base;
  uint32_t cur, nxt, cidx, wpos;
  uint64_t bb;
  int bc;
};
// start reading at bit `bitpos` counted from the 4-byte aligned pointer `base`
__device__ __forceinline__ void ub_init(UBits& s, const uint32_t* base, uint64_t bitpos, int lane) {
  s.base = (v4_gsrc_t)base;
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

#ifdef V4_GUARD
__device__ unsigned int v4_guard_word[8];
#define V4_G(cond, code, val) ((cond) ? (atomicOr(&v4_guard_word[0], 1u << (code)), atomicMax(&v4_guard_word[code], (unsigned)(val)), true) : false)
#else
#define V4_G(cond, code, val) false
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
// the entry of symbol `sym` reached with `code_bits` code bits left to consume at this lookup
__device__ __forceinline__ uint32_t sym_entry(int sym, int code_bits, bool is_dist) {
  uint32_t base, eb;
  if (is_dist) {
    if (sym > 29) return E4_BAD;
    dist_base_extra((uint32_t)sym, &base, &eb);
    return v4_enc(V4_LIT_BITS, V4_LIT_ROOT, (uint32_t)code_bits + eb, eb, (base - 1u) >> eb);
  }
  if (sym < 256) return v4_enc(V4_LIT_BITS, V4_LIT_ROOT, (uint32_t)code_bits, 0u, (uint32_t)sym);
  if (sym == 256) return v4_enc(0u, V4_STOP_EOB, (uint32_t)code_bits, 0u, 0u);
  if (sym > 285) return E4_BAD;
  len_base_extra((uint32_t)(sym - 257), &base, &eb);
  return v4_enc(V4_DIST_BITS, V4_DIST_ROOT, (uint32_t)code_bits + eb, eb, (base - 3u) >> eb);
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
// `root`: entry offset of the root table.  Sub-tables: the literal/length alphabet takes them from `sub_hi` downwards, the
// distance alphabet (one chunk of symbols, so the total is known at once) from `sub_lo` upwards; `sub_total` = entries
// taken.  Returns 1 for an invalid code, 2 when the sub-tables do not fit [sub_lo, sub_hi).
__device__ __forceinline__ int v4_build(V4Lds& L, const uint8_t* lens, int n, uint32_t root, int root_bits, uint32_t sub_lo, uint32_t sub_hi,
                        uint16_t* sorted, bool is_dist, int lane, uint32_t& sub_total) {
  uint32_t* const fast = L.tab;
  sub_total = 0;
  V4_SYNC();
  for (int i = lane; i < (1 << root_bits); i += WAVE) fast[root + i] = E4_BAD;  // bit patterns no code maps to
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
  V4_SYNC();
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
      const uint32_t tot_here = __builtin_amdgcn_readlane(suf, 0);  // everything from this chunk's first group to the right
      if (tot_here > sub_hi - sub_lo) { over = 2; break; }
      // entry offset of the group's sub-table (distance alphabet: its codes longer than the root are < 64, one chunk)
      const uint32_t base = is_dist ? sub_lo + tot_here - suf : sub_hi - suf;
      sub_total = tot_here;
      // (every sub-table size is a power of two >= 2 and the space ends on an even entry, so `base` is even)
      if (is_tail) fast[root + bitrev2(prefix, root_bits)] = v4_enc(sbits, base, (uint32_t)root_bits, 0u, 0u);  // following the pointer consumes the root bits
      if (in) {
        const uint32_t r = bitrev2(code, (int)len) >> root_bits;  // bits after the root, LSB-first
        const uint32_t e = sym_entry((int)sym, (int)sub_len, is_dist);
        for (uint32_t i = r; i < (1u << sbits); i += (1u << sub_len)) fast[base + i] = e;
      }
      carry_prefix = __builtin_amdgcn_readlane(prefix, 0);
      carry_sbits = __builtin_amdgcn_readlane(sbits, 0);
      carry_suffix = __builtin_amdgcn_readlane(suf, 0);
    }
    if (over) return over;
  }
  // 5. root entries: one symbol per lane, replicated over the unused high index bits
  for (uint32_t k = lane; k < k0; k += WAVE) {
    const int sym = sorted[k];
    const int l = lens[sym];
    const uint32_t c = (uint32_t)L.b.t_first[l] + (k - L.b.t_offs[l]);
    const uint32_t r = bitrev2(c, l);
    const uint32_t e = sym_entry(sym, l, is_dist);
    for (uint32_t i = r; i < (1u << root_bits); i += (1u << l)) fast[root + i] = e;
  }
  V4_SYNC();
  return 0;
}

// ---- the decode step ---------------------------------------------------------------------------------
// One table lookup per loop iteration.  A lane's bit window is 32 bits starting at `pos`, funnel-shifted (v_alignbit) out of
// two input dwords d0, d1; `nxt` is the dword behind them, prefetched; `xc` is the bit position at which the window crosses
// into d1.  A step consumes <= 22 bits, so pos crosses at most one dword per step.  The lane's state is (tb, st): the entry
// offset of the table its next lookup goes to and a register whose low bits hold that table's index width -- the entry
// read last.  tb == 0 is the literal/length root, i.e. a symbol boundary.
//   w   = alignbit(d1, d0, pos)          e   = tab[tb + bfe(w, 0, st)]
//   pos += e.adv                         tb  = e.ntb, st = e
// A lane leaves the loop (EXEC) when it stands on a symbol boundary at or behind its limit, or when it is sent to a STOP
// table; what it was doing is read from its registers afterwards.
// The compressed input is read through an address-space-1 pointer: `base32` is made from an integer (alignment of the payload
// pointer), which hides from the compiler that it points to global memory, and a generic pointer is read with FLAT loads.
// A FLAT load counts in lgkmcnt as well as in vmcnt, so the wait behind the table lookup would wait for the bit-window
// prefetch as well.
#ifdef V4_GUARD
#define V4_SRC(i) (V4_G((i) >= V4_STAGE_DW, 1, (i)) ? 0u : L.stage[i])
#else
#define V4_SRC(i) L.stage[i]
#endif
#define V4_WIN_DECL(p0) uint32_t d0 = V4_SRC((p0) >> 5), d1 = V4_SRC(((p0) >> 5) + 1u), nxt = V4_SRC(((p0) >> 5) + 2u), xc = ((p0) & ~31u) + 32u
// explicit moves: left to the register allocator, the fresh load is copied into place right away and the
// wave waits for it here instead of one crossing later
#define V4_WIN_CROSS()                                                                                   \
  do {                                                                                                   \
    asm volatile("v_mov_b32 %0, %1" : "=v"(d0) : "v"(d1));                                              \
    asm volatile("v_mov_b32 %0, %1" : "=v"(d1) : "v"(nxt));                                             \
    xc += 32u;                                                                                           \
    nxt = V4_SRC((xc >> 5) + 1u);                                                                        \
  } while (0)
#define V4_ADV(e) __builtin_amdgcn_ubfe((e), 15u, 5u)
#define V4_NTB(e) __builtin_amdgcn_ubfe((e), 4u, 11u)
// the length (minus 3) / distance (minus 1) a symbol entry and the window it was looked up in stand for
__device__ __forceinline__ uint32_t v4_value(uint32_t e, uint32_t w, uint32_t adv) {
  const uint32_t eb = __builtin_amdgcn_ubfe(e, 20u, 4u);
  return ((e >> 24) << eb) + __builtin_amdgcn_ubfe(w, adv - eb, eb);
}
constexpr uint32_t V4_CK_ROW = 3u * 64u;  // dwords per checkpoint index: pos[64], acc[64], state[64]
static_assert((uint32_t)V4_CK_MAX * V4_CK_ROW == V4_CK_DWORDS, "kernels.h sizes the checkpoint scratch for V4_CK_MAX rows");

// SYNC pass: a speculative lane decodes from `start` (an arbitrary bit, `ov` bits in front of its boundary) only to find the
// symbol chain: it stops at the first symbol start at or after `count_from` and reports it.  Nothing is counted.  Garbage
// contains END-OF-BLOCK and unassigned codes: such a stop says nothing about the block, the STOP tables lead the lane back
// to the literal/length root (13 vector instructions per step; v3: 33).
__device__ __forceinline__ uint32_t v4_sync(V4Lds& L, uint32_t start, uint32_t count_from, uint32_t limit, v4_gsrc_t gsrc) {
  uint32_t pos = start;
  const bool run0 = pos < count_from;  // a lane that starts exactly on its boundary (lane 0: the round's first bit) is there already
  const uint32_t p0 = run0 ? pos : 0u;
  V4_WIN_DECL(p0);
  uint32_t tb = V4_LIT_ROOT, st = (uint32_t)V4_LIT_BITS;
  bool run = run0;
  while (__ballot(run) != 0ull) {
    if (run) {
#ifdef V4_ASM_MARKERS
      asm volatile("; V4LOOP_BEGIN %0" ::"n"(3));
#endif
      const uint32_t w = __builtin_amdgcn_alignbit(d1, d0, pos);
      const uint32_t e = L.tab[tb + __builtin_amdgcn_ubfe(w, 0u, st)];
      pos += V4_ADV(e);
      tb = V4_NTB(e);
      st = e;
      if (pos >= xc) V4_WIN_CROSS();
      run = !(tb == V4_LIT_ROOT && pos >= count_from);
#ifdef V4_ASM_MARKERS
      asm volatile("; V4LOOP_END %0" ::"n"(3));
#endif
    }
  }
  (void)limit;
  return pos;
}

// COUNT pass of this lane's sub-stream [start, limit), `start` being a symbol start: bytes produced [0:19] and matches
// [20:31] in one accumulator.  Every V4_CK_STEPS steps
// (wave-uniform counter) a lane that is still running stores (pos, acc, tb | index width << 12 | pending match length << 16)
// to checkpoint row c of the wave's scratch; ck_n = number of the lane's valid rows (rows 1 .. ck_n).  Returns true
// (uniform) when a pass needs more than V4_CK_MAX rows: the caller restarts the round with short sub-streams.
#ifdef V4_UTIL
__device__ unsigned long long v4_util[4];  // dev diagnostic: loop iterations / lane-steps of the first count pass, of the fix passes
#endif
__device__ __forceinline__ bool v4_count(V4Lds& L, bool active, uint32_t start, uint32_t limit,
                                         v4_gsrc_t gsrc, uint32_t* __restrict__ ck, int lane,
                                         uint32_t& end_out, uint32_t& acc_out, uint32_t& flags, uint32_t& ck_n, int kind = 0) {
  uint32_t pos = start;
  uint32_t acc = 0;
  const bool run0 = active && pos < limit;
  // (a lane that does not run never loads again: its three reads are parked at 0)
  const uint32_t p0 = run0 ? pos : 0u;
  V4_WIN_DECL(p0);
  uint32_t tb = V4_LIT_ROOT, st = (uint32_t)V4_LIT_BITS, mlen = 0;
  const uint32_t lit_lo = uni2(L.lit_lo);
  uint32_t cd = V4_CK_STEPS, c = 0;  // wave-uniform: steps to the next checkpoint, checkpoint row
  uint32_t cn = 0;
  bool overflow = false;
  bool run = run0;
#ifdef V4_UTIL
  uint32_t u_it = 0, u_act = 0;
#endif
  while (__ballot(run) != 0ull) {
#ifdef V4_UTIL
    u_it++; u_act += (uint32_t)__popcll(__ballot(run));
#endif
    if (cd == 0) {
      cd = V4_CK_STEPS;
      c++;
      if (c >= (uint32_t)V4_CK_MAX) { overflow = true; break; }
      if (run) {
        uint32_t* q = ck + c * V4_CK_ROW + (uint32_t)lane;
        q[0] = pos; q[64] = acc; q[128] = tb | ((st & 15u) << 12) | ((mlen & 0x1FFu) << 16);
        cn = c;
      }
    }
    cd--;
    if (run) {
#ifdef V4_ASM_MARKERS
      asm volatile("; V4LOOP_BEGIN %0" ::"n"(0));
#endif
      const bool in_lit = tb >= lit_lo;
      const uint32_t w = __builtin_amdgcn_alignbit(d1, d0, pos);
      const uint32_t e = L.tab[tb + __builtin_amdgcn_ubfe(w, 0u, st)];
      const uint32_t adv = V4_ADV(e);
      pos += adv;
      tb = V4_NTB(e);
      st = e;
      if (pos >= xc) V4_WIN_CROSS();
      const bool is_len = tb == V4_DIST_ROOT;             // only a length symbol leads to the distance root
      const bool done = tb == V4_LIT_ROOT;                // a literal or a distance symbol: back at a symbol boundary
      if (is_len) mlen = v4_value(e, w, adv) + (3u + (1u << 20));   // the match count rides in the same accumulator
      acc += done ? (in_lit ? 1u : mlen) : 0u;               // a match is counted where it is written: at its distance symbol
      run = !(done && pos >= limit) && tb < V4_STOP_EOB;
#ifdef V4_ASM_MARKERS
      asm volatile("; V4LOOP_END %0" ::"n"(0));
#endif
    }
  }
#ifdef V4_UTIL
  if (lane == 0) { atomicAdd(&v4_util[2 * kind], (unsigned long long)u_it); atomicAdd(&v4_util[2 * kind + 1], (unsigned long long)u_act); }
#endif
  if (active) {
    end_out = pos;
    acc_out = acc; ck_n = cn;
    flags = tb == V4_STOP_EOB ? F_EOB : (tb == V4_STOP_BAD ? F_BAD : 0u);
  }
  return overflow;
}

// ---- the sync and count loops, written out (persistent launch shape: the tables start at LDS address 0) --------------
// Compiled from the C++ above the two loops cost 19 and 48 vector + 19 and 45 scalar instructions per step: the loop-carried
// `run` predicate, the phi copies at the loop head and the EXEC bookkeeping of two nested `if`s.  A wave's own instruction
// stream (every kind of instruction, about one per five cycles) bounds K1 as much as the vector ALUs do (4-5 waves per
// SIMD), so the loops are written with the instructions they need and nothing else:
// sync 13 vector + 7 scalar, count 25 vector + 8 scalar per step.  EXEC holds the lanes that are still running; the
// block ends with the lanes that survived its steps in `run`.  Hazards the assembler does not pad inside inline asm: a
// vector compare that writes VCC / an SGPR pair is never read as a VALU mask by one of the next two instructions.
#define V4_ASM_CROSS                                                            \
    "v_cmp_le_u32 vcc, %[xc], %[pos]\n\t"                                       \
    "s_and_saveexec_b64 %[s1], vcc\n\t"                                         \
    "v_mov_b32 %[d0], %[d1]\n\t"                                                \
    "v_add_u32 %[xc], 32, %[xc]\n\t"                                            \
    "v_lshrrev_b32 %[t], 3, %[xc]\n\t"                                          \
    "v_mov_b32 %[d1], %[nxt]\n\t"   /* (its read was issued a step ago at the least: the table lookup's wait covered it) */ \
    "ds_read_b32 %[nxt], %[t] offset:%[soff]\n\t"                               \
    "s_or_b64 exec, exec, %[s1]\n\t"
__device__ __forceinline__ uint32_t v4_sync_asm(V4Lds& L, uint32_t start, uint32_t count_from) {
  uint32_t pos = start;
  const bool run0 = pos < count_from;
  const uint32_t p0 = run0 ? pos : 0u;
  uint32_t d0 = L.stage[p0 >> 5], d1 = L.stage[(p0 >> 5) + 1u], nxt = L.stage[(p0 >> 5) + 2u], xc = (p0 & ~31u) + 32u;
  uint32_t tb = V4_LIT_ROOT, st = (uint32_t)V4_LIT_BITS, w, t;
  const unsigned long long run = __ballot(run0);
  unsigned long long sav, s1;
  asm volatile(
    "s_mov_b64 %[sav], exec\n\t"
    "s_and_b64 exec, exec, %[run]\n\t"
    "s_cbranch_execz 2f\n"
    "1:\n\t"
    "v_alignbit_b32 %[w], %[d1], %[d0], %[pos]\n\t"
    "v_bfe_u32 %[t], %[w], 0, %[st]\n\t"
    "v_add_lshl_u32 %[t], %[t], %[tb], 2\n\t"
    "ds_read_b32 %[st], %[t]\n\t"
    "s_waitcnt lgkmcnt(0)\n\t"
    "v_bfe_u32 %[t], %[st], 15, 5\n\t"
    "v_add_u32 %[pos], %[pos], %[t]\n\t"
    "v_bfe_u32 %[tb], %[st], 4, 11\n\t"
    V4_ASM_CROSS
    "v_cmp_ne_u32 vcc, %[clit], %[tb]\n\t"        // not on a symbol boundary
    "v_cmp_gt_u32 %[s1], %[cf], %[pos]\n\t"       // or in front of count_from: carry on
    "s_or_b64 vcc, vcc, %[s1]\n\t"
    "s_and_b64 exec, exec, vcc\n\t"
    "s_cbranch_execnz 1b\n"
    "2:\n\t"
    "s_mov_b64 exec, %[sav]\n\t"
    "s_waitcnt lgkmcnt(0)"
    : [pos] "+v"(pos), [tb] "+v"(tb), [st] "+v"(st), [d0] "+v"(d0), [d1] "+v"(d1), [nxt] "+v"(nxt), [xc] "+v"(xc),
      [w] "=&v"(w), [t] "=&v"(t), [sav] "=&s"(sav), [s1] "=&s"(s1)
    : [run] "s"(run), [cf] "v"(count_from), [clit] "s"(V4_LIT_ROOT), [soff] "n"(offsetof(V4Lds, stage) + 4)
    : "vcc", "scc", "memory");
  return pos;
}

// One block of at most V4_CK_STEPS count steps for the lanes in `run` (see v4_count).
__device__ __forceinline__ void v4_count_asm_block(unsigned long long& run, uint32_t& pos, uint32_t& tb, uint32_t& st, uint32_t& d0, uint32_t& d1,
                                                   uint32_t& nxt, uint32_t& xc, uint32_t& acc, uint32_t& mlen, uint32_t limit, uint32_t lit_lo) {
  uint32_t w, t, a, k;
  unsigned long long sav, s1, s2, s3;
  asm volatile(
    "s_mov_b64 %[sav], exec\n\t"
    "s_and_b64 exec, exec, %[run]\n\t"
    "s_cbranch_execz 2f\n\t"
    "s_movk_i32 %[k], %[ksteps]\n"
    "1:\n\t"
    "v_alignbit_b32 %[w], %[d1], %[d0], %[pos]\n\t"
    "v_cmp_le_u32 %[s3], %[clo], %[tb]\n\t"       // this lookup is in a literal/length table
    "v_bfe_u32 %[t], %[w], 0, %[st]\n\t"
    "v_add_lshl_u32 %[t], %[t], %[tb], 2\n\t"
    "ds_read_b32 %[st], %[t]\n\t"
    "s_waitcnt lgkmcnt(0)\n\t"
    "v_bfe_u32 %[a], %[st], 15, 5\n\t"
    "v_add_u32 %[pos], %[pos], %[a]\n\t"
    "v_bfe_u32 %[tb], %[st], 4, 11\n\t"
    V4_ASM_CROSS
    "v_bfe_u32 %[t], %[st], 20, 4\n\t"            // eb
    "v_cmp_eq_u32 %[s2], 0, %[tb]\n\t"            // a length symbol: the next lookup is the distance root
    "v_sub_u32 %[a], %[a], %[t]\n\t"              // code bits
    "v_cmp_eq_u32 vcc, %[clit], %[tb]\n\t"        // a literal or a distance symbol: back on a symbol boundary
    "v_bfe_u32 %[a], %[w], %[a], %[t]\n\t"        // extra bits
    "v_lshlrev_b32_sdwa %[t], %[t], %[st] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"  // m << eb
    "v_add3_u32 %[t], %[t], %[a], %[clen]\n\t"    // length + (1 << 20): the match count rides in the same accumulator
    "v_cndmask_b32 %[mlen], %[mlen], %[t], %[s2]\n\t"
    "v_cndmask_b32 %[t], %[mlen], 1, %[s3]\n\t"   // what a completed symbol produces: 1 byte, or the pending match
    "v_cmp_ge_u32 %[s1], %[pos], %[limit]\n\t"
    "v_cndmask_b32 %[t], 0, %[t], vcc\n\t"
    "v_cmp_gt_u32 %[s2], %[cstop], %[tb]\n\t"     // not sent to a STOP table
    "v_add_u32 %[acc], %[acc], %[t]\n\t"
    "s_and_b64 vcc, vcc, %[s1]\n\t"               // on a symbol boundary at or behind the limit: done
    "s_andn2_b64 %[s1], %[s2], vcc\n\t"
    "s_and_b64 exec, exec, %[s1]\n\t"
    "s_cbranch_execz 2f\n\t"
    "s_sub_u32 %[k], %[k], 1\n\t"
    "s_cbranch_scc0 1b\n"
    "2:\n\t"
    "s_mov_b64 %[run], exec\n\t"
    "s_mov_b64 exec, %[sav]\n\t"
    "s_waitcnt lgkmcnt(0)"
    : [run] "+s"(run), [pos] "+v"(pos), [tb] "+v"(tb), [st] "+v"(st), [d0] "+v"(d0), [d1] "+v"(d1), [nxt] "+v"(nxt), [xc] "+v"(xc),
      [acc] "+v"(acc), [mlen] "+v"(mlen), [w] "=&v"(w), [t] "=&v"(t), [a] "=&v"(a), [k] "=&s"(k),
      [sav] "=&s"(sav), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3)
    : [limit] "v"(limit), [soff] "n"(offsetof(V4Lds, stage) + 4), [clo] "s"(lit_lo), [clit] "s"(V4_LIT_ROOT), [cstop] "s"(V4_STOP_EOB), [clen] "s"(3u + (1u << 20)),
      [ksteps] "n"(V4_CK_STEPS - 1)
    : "vcc", "scc", "memory");
}

__device__ __forceinline__ bool v4_count_asm(V4Lds& L, bool active, uint32_t start, uint32_t limit, uint32_t* __restrict__ ck, int lane,
                                             uint32_t& end_out, uint32_t& acc_out, uint32_t& flags, uint32_t& ck_n) {
  uint32_t pos = start, acc = 0;
  const bool run0 = active && pos < limit;
  const uint32_t p0 = run0 ? pos : 0u;
  uint32_t d0 = L.stage[p0 >> 5], d1 = L.stage[(p0 >> 5) + 1u], nxt = L.stage[(p0 >> 5) + 2u], xc = (p0 & ~31u) + 32u;
  uint32_t tb = V4_LIT_ROOT, st = (uint32_t)V4_LIT_BITS, mlen = 0;
  const uint32_t lit_lo = uni2(L.lit_lo);
  uint32_t c = 0, cn = 0;
  bool overflow = false;
  unsigned long long run = __ballot(run0);
  while (run != 0ull) {
    v4_count_asm_block(run, pos, tb, st, d0, d1, nxt, xc, acc, mlen, limit, lit_lo);
    // (an SGPR result of inline asm counts as divergent; said to be uniform, the loops around this one stay scalar)
    run = (unsigned long long)uni2((uint32_t)run) | (unsigned long long)uni2((uint32_t)(run >> 32)) << 32;
    if (run == 0ull) break;
    c++;
    if (c >= (uint32_t)V4_CK_MAX) { overflow = true; break; }
    if ((run >> lane) & 1ull) {
      uint32_t* q = ck + c * V4_CK_ROW + (uint32_t)lane;
      q[0] = pos; q[64] = acc; q[128] = tb | ((st & 15u) << 12) | ((mlen & 0x1FFu) << 16);
      cn = c;
    }
  }
  if (active) {
    end_out = pos;
    acc_out = acc; ck_n = cn;
    flags = tb == V4_STOP_EOB ? F_EOB : (tb == V4_STOP_BAD ? F_BAD : 0u);
  }
  return overflow;
}

// WRITE pass of one SEGMENT: the lane resumes the decoder state (pos, tb, index width, pending match length) of a
// checkpoint (or the exact start of a sub-stream: root state) and decodes until pos reaches `stop_any` (the next checkpoint
// of that sub-stream: reached exactly, in whatever state) or, for a sub-stream's last segment, its ordinary end (first
// symbol start at / after `limit`, or END-OF-BLOCK).  MODE 1: literals to HBM; 2: literals to the LDS window (window byte
// 0 = output byte win_base).  Matches are appended to `mlist` at mpos.  Returns F_BAD when a distance reaches before the
// output's start.
template <int MODE>
__device__ __forceinline__ uint32_t v4_write(V4Lds& L, bool active, uint32_t pos, uint32_t tb0, uint32_t mb0, uint32_t mlen,
                                             uint32_t limit, uint32_t stop_any, uint8_t* out, uint32_t opos,
                                             unsigned long long* mlist, uint32_t mpos, uint32_t win_base,
                                             v4_gsrc_t gsrc) {
  const bool run0 = active && pos < stop_any;
  const uint32_t p0 = run0 ? pos : 0u;
  V4_WIN_DECL(p0);
  uint32_t tb = tb0, st = mb0;
  const uint32_t lit_lo = uni2(L.lit_lo);
  bool run = run0, bad = false;
  while (__ballot(run) != 0ull) {
    if (run) {
#ifdef V4_ASM_MARKERS
      asm volatile("; V4LOOP_BEGIN %0" ::"n"(MODE));
#endif
      const bool in_lit = tb >= lit_lo;
      const uint32_t w = __builtin_amdgcn_alignbit(d1, d0, pos);
      const uint32_t e = L.tab[tb + __builtin_amdgcn_ubfe(w, 0u, st)];
      const uint32_t adv = V4_ADV(e);
      pos += adv;
      tb = V4_NTB(e);
      st = e;
      if (pos >= xc) V4_WIN_CROSS();
      const bool is_len = tb == V4_DIST_ROOT;
      const bool done = tb == V4_LIT_ROOT;
      const bool is_lit = done && in_lit, is_dist = done && !in_lit;
      const uint32_t val = v4_value(e, w, adv);
      if (MODE == 1) { if (is_lit && !V4_G(opos >= win_base, 2, opos)) out[opos] = (uint8_t)(e >> 24); }
      if (MODE == 2) { if (is_lit) (L.win - win_base)[opos] = (uint8_t)(e >> 24); }  // base pointer folded: one VALU less than an index subtraction
      if (is_len) mlen = val + 3u;
      bool okm = is_dist;
      if (okm && val >= opos) { bad = true; okm = false; }   // distance val + 1 reaches before the output's start
      if (MODE == 1) { if (okm && !V4_G(mpos >= V4_ML_ENTRIES, 3, mpos)) { uint2 ent; ent.x = opos; ent.y = mlen | ((val + 1u) << 12); ((uint2*)mlist)[mpos] = ent; } }  // = opos | mlen << 32 | dist << 44
      if (MODE == 2) {
        if (okm && !V4_G(mpos >= V4_LCAP, 3, mpos)) {
          uint8_t* q = (L.win - win_base) + opos;
          L.ml16[mpos] = (uint16_t)(opos - win_base);
          q[0] = (uint8_t)(mlen - 3u); q[1] = (uint8_t)val; q[2] = (uint8_t)(val >> 8);
        }
      }
      mpos += okm ? 1u : 0u;
      opos += is_lit ? 1u : (okm ? mlen : 0u);
      run = !(done && pos >= limit) && tb < V4_STOP_EOB && pos < stop_any && !bad;
#ifdef V4_ASM_MARKERS
      asm volatile("; V4LOOP_END %0" ::"n"(MODE));
#endif
    }
  }
  return (bad || tb == V4_STOP_BAD) ? F_BAD : 0u;
}


// The write loop of a segment whose output lands in the LDS window (v4_write<2>, written out: 30 vector + 14 scalar
// instructions per step against 43 + 60 from the compiler).  `orel` is the lane's output position relative to the window's
// first byte R; a match goes to the on-chip list (V4Lds::ml16, byte offset mp8 = 2 x its number in the mini-round).
// The loop has no wait but the table lookup's: the compressed bits come from LDS and so does everything it writes.
__device__ __forceinline__ uint32_t v4_write_win_asm(V4Lds& L, bool active, uint32_t pos, uint32_t tb, uint32_t st, uint32_t mlen,
                                                     uint32_t limit, uint32_t stop_any, uint32_t orel, uint32_t R, uint32_t mp8) {
  const bool run0 = active && pos < stop_any;
  const uint32_t p0 = run0 ? pos : 0u;
  uint32_t d0 = L.stage[p0 >> 5], d1 = L.stage[(p0 >> 5) + 1u], nxt = L.stage[(p0 >> 5) + 2u], xc = (p0 & ~31u) + 32u;
  uint32_t w, t, a;
  const unsigned long long run = __ballot(run0);
  unsigned long long sav, s1, s2, s3, s4, sbad = 0;
  R = uni2(R);   // (said to be uniform: an "s" operand the compiler takes for divergent is handed over in vector registers)
  const uint32_t lit_lo = uni2(L.lit_lo);
  asm volatile(
    "s_mov_b64 %[sav], exec\n\t"
    "s_and_b64 exec, exec, %[run]\n\t"
    "s_cbranch_execz 2f\n"
    "1:\n\t"
    "v_alignbit_b32 %[w], %[d1], %[d0], %[pos]\n\t"
    "v_cmp_le_u32 %[s3], %[clo], %[tb]\n\t"       // this lookup is in a literal/length table
    "v_bfe_u32 %[t], %[w], 0, %[st]\n\t"
    "v_add_lshl_u32 %[t], %[t], %[tb], 2\n\t"
    "ds_read_b32 %[st], %[t]\n\t"
    "s_waitcnt lgkmcnt(0)\n\t"
    "v_bfe_u32 %[a], %[st], 15, 5\n\t"
    "v_add_u32 %[pos], %[pos], %[a]\n\t"
    "v_bfe_u32 %[tb], %[st], 4, 11\n\t"
    V4_ASM_CROSS
    "v_bfe_u32 %[t], %[st], 20, 4\n\t"            // eb
    "v_cmp_eq_u32 %[s2], 0, %[tb]\n\t"            // a length symbol
    "v_sub_u32 %[a], %[a], %[t]\n\t"              // code bits
    "v_cmp_eq_u32 %[s4], %[clit], %[tb]\n\t"      // a literal or a distance symbol: back on a symbol boundary
    "v_bfe_u32 %[a], %[w], %[a], %[t]\n\t"        // extra bits
    "v_lshlrev_b32_sdwa %[t], %[t], %[st] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"  // m << eb
    "v_add3_u32 %[t], %[t], %[a], 3\n\t"          // a length (a distance + 2)
    "v_cndmask_b32 %[mlen], %[mlen], %[t], %[s2]\n\t"
    "s_and_b64 %[s2], %[s4], %[s3]\n\t"           // literal
    "s_andn2_b64 %[s3], %[s4], %[s3]\n\t"         // distance symbol: a match is complete
    "s_and_saveexec_b64 %[s1], %[s2]\n\t"
    "v_lshrrev_b32 %[a], 24, %[st]\n\t"
    "ds_write_b8 %[orel], %[a] offset:%[woff]\n\t"
    "v_add_u32 %[orel], 1, %[orel]\n\t"
    "s_mov_b64 exec, %[s1]\n\t"
    "s_and_saveexec_b64 %[s1], %[s3]\n\t"
    "v_add_u32 %[a], -3, %[t]\n\t"                // the distance - 1
    "v_add_u32 %[w], %[cr], %[orel]\n\t"          // the absolute output position
    "v_cmp_ge_u32 vcc, %[a], %[w]\n\t"            // reaches before the output's start
    "s_or_b64 %[sbad], %[sbad], vcc\n\t"
    "s_andn2_b64 exec, exec, vcc\n\t"
    "v_add_u32 %[w], -3, %[mlen]\n\t"
    "ds_write_b16 %[mp8], %[orel] offset:%[loff]\n\t"     // where the match is ...
    "ds_write_b8 %[orel], %[w] offset:%[woff]\n\t"        // ... and what it is, in its own first three bytes
    "v_lshrrev_b32 %[w], 8, %[a]\n\t"
    "ds_write_b8 %[orel], %[a] offset:%[woff1]\n\t"
    "ds_write_b8 %[orel], %[w] offset:%[woff2]\n\t"
    "v_add_u32 %[mp8], 2, %[mp8]\n\t"
    "v_add_u32 %[orel], %[orel], %[mlen]\n\t"
    "s_mov_b64 exec, %[s1]\n\t"
    "v_cmp_ge_u32 %[s1], %[pos], %[limit]\n\t"
    "v_cmp_gt_u32 %[s2], %[cstop], %[tb]\n\t"     // not sent to a STOP table
    "v_cmp_gt_u32 %[s3], %[stopany], %[pos]\n\t"  // the next checkpoint not reached
    "s_and_b64 %[s1], %[s1], %[s4]\n\t"           // on a symbol boundary at or behind the limit
    "s_andn2_b64 %[s2], %[s2], %[s1]\n\t"
    "s_andn2_b64 %[s3], %[s3], %[sbad]\n\t"
    "s_and_b64 %[s2], %[s2], %[s3]\n\t"
    "s_and_b64 exec, exec, %[s2]\n\t"
    "s_cbranch_execnz 1b\n"
    "2:\n\t"
    "s_mov_b64 exec, %[sav]\n\t"
    "s_waitcnt lgkmcnt(0)"
    : [pos] "+v"(pos), [tb] "+v"(tb), [st] "+v"(st), [d0] "+v"(d0), [d1] "+v"(d1), [nxt] "+v"(nxt), [xc] "+v"(xc),
      [mlen] "+v"(mlen), [orel] "+v"(orel), [mp8] "+v"(mp8), [sbad] "+s"(sbad), [w] "=&v"(w), [t] "=&v"(t), [a] "=&v"(a),
      [sav] "=&s"(sav), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3), [s4] "=&s"(s4)
    : [run] "s"(run), [limit] "v"(limit), [stopany] "v"(stop_any), [cr] "s"(R), [clo] "s"(lit_lo), [clit] "s"(V4_LIT_ROOT),
      [cstop] "s"(V4_STOP_EOB), [soff] "n"(offsetof(V4Lds, stage) + 4), [woff] "n"(offsetof(V4Lds, win)),
      [woff1] "n"(offsetof(V4Lds, win) + 1), [woff2] "n"(offsetof(V4Lds, win) + 2), [loff] "n"(offsetof(V4Lds, ml16))
    : "vcc", "scc", "memory");
  const bool bad = ((sbad >> (threadIdx.x & 63)) & 1ull) != 0ull;
  return (bad || tb == V4_STOP_BAD) ? F_BAD : 0u;
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
__device__ __forceinline__ void v4_resolve_batch(uint8_t* out, int lane, int nm, uint32_t m_dst, uint32_t m_len, uint32_t m_dist) {
  bool valid = lane < nm;
  if (valid && V4_G(m_dst + m_len > 65536u || m_dist > m_dst || m_len > 258u, 4, m_dst + m_len)) valid = false;
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
// of its source that precedes R (v4_far_copy) and the matches that also read the window are compacted to the front
// of the list, then only those go through the dependency-ordered copy (v4_near_batch), in dense batches of 64.
// v4_far_issue loads the first 16 bytes of the far part (sources before R are final bytes in HBM); v4_far_finish stores them
// into the window and copies what lies beyond 16 bytes.  Split in two so that the caller can have the next batch's loads
// in flight while it stores this batch's (the load latency was the largest single wait of the resolve).
__device__ __forceinline__ uint32_t v4_far_count(uint32_t R, bool valid, uint32_t m_dst, uint32_t m_len, uint32_t m_dist) {
  const uint32_t src_lo = m_dst - m_dist;
  uint32_t n_far = 0;
  if (valid && src_lo < R) { n_far = R - src_lo; if (n_far > m_len) n_far = m_len; }
  return n_far;
}
__device__ __forceinline__ u32x4 v4_far_issue(const uint8_t* out, uint32_t n_far, uint32_t m_dst, uint32_t m_dist) {
  u32x4 v = {0, 0, 0, 0};
  if (n_far) v = ld16(out + (m_dst - m_dist));  // over-read is inside the (padded) buffer
  return v;
}
__device__ __forceinline__ void v4_far_finish(uint8_t* win, const uint8_t* out, uint32_t R, uint32_t n_far, u32x4 v, uint32_t m_dst, uint32_t m_dist) {
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
__device__ __forceinline__ void v4_near_batch(uint8_t* win, uint32_t R, int lane, int nm, uint32_t m_dst, uint32_t m_len, uint32_t m_dist) {
  bool valid = lane < nm;
  if (valid && V4_G(m_dst + m_len > 65536u || m_dist > m_dst || m_len > 258u || m_dst < R || m_dst + m_len - R > (uint32_t)V4_WIN, 5, m_dst + m_len)) valid = false;
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
  uint32_t n_far = 0;  // already copied by v4_far_copy
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
// use_win: the range lives in the LDS window (two walks, see v4_far_copy / v4_near_batch) and is flushed to HBM with
// coalesced 16-byte stores afterwards; otherwise literals are already in HBM and the matches are copied there.
__device__ __forceinline__ void v4_resolve(V4Lds& L, uint8_t* out, unsigned long long* mlist, int lane, uint32_t R, uint32_t tot_out,
                                           uint32_t tot_m, bool use_win, bool dbg, uint32_t& dbg_matches, uint32_t& dbg_near, unsigned long long* tcx) {
  unsigned long long tr0 = dbg ? clock64() : 0;
#define RTOCK(i) do { if (dbg) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); unsigned long long t1 = clock64(); tcx[i] += t1 - tr0; tr0 = t1; } } while (0)  /* (the anatomy run drains the memory pipeline at every mark: what a phase waits for is charged to it) */
  unsigned long long m_next = 0;
  if (!use_win && (uint32_t)lane < tot_m) m_next = mlist[lane];
  if (use_win) {
    // entry k of the on-chip list as destination | length << 32 | distance << 44 (0 behind the end)
    auto entry = [&](uint32_t k) -> unsigned long long {
      if (k >= tot_m) return 0ull;
      const uint32_t rel = L.ml16[k];
      const uint32_t h = ((const u32p*)(L.win + rel))->v;   // length - 3 | distance - 1 << 8 (the fourth byte is somebody else's)
      return (unsigned long long)(R + rel) | (unsigned long long)((h & 0xFFu) + 3u) << 32 | (unsigned long long)(((h >> 8) & 0x7FFFu) + 1u) << 44;
    };
    uint32_t n_near = 0;  // matches that also read this round's window, compacted to the front of the wave's list in global memory
    // software pipeline: batch k + 1's far source is loaded before batch k's bytes are stored.  (Every entry is read from the
    // window before the copy of ITS match overwrites it: a batch's entries are in registers one batch ahead.)
    unsigned long long m = entry((uint32_t)lane);
    m_next = entry(WAVE + (uint32_t)lane);
    uint32_t md = (uint32_t)(m & 0xFFFFFFFFull), ml = (uint32_t)((m >> 32) & 0xFFFu), mdist = (uint32_t)(m >> 44);
    uint32_t nf = v4_far_count(R, (uint32_t)lane < tot_m, md, ml, mdist);
    u32x4 fv = v4_far_issue(out, nf, md, mdist);
    for (uint32_t k = 0; k < tot_m; k += WAVE) {
      const uint32_t nmb = tot_m - k < WAVE ? tot_m - k : WAVE;
      const bool valid_m = (uint32_t)lane < nmb;
      const unsigned long long m2 = m_next;
      m_next = entry(k + 2 * WAVE + (uint32_t)lane);
      const uint32_t md2 = (uint32_t)(m2 & 0xFFFFFFFFull), ml2 = (uint32_t)((m2 >> 32) & 0xFFFu), mdist2 = (uint32_t)(m2 >> 44);
      const uint32_t nf2 = v4_far_count(R, k + WAVE + (uint32_t)lane < tot_m, md2, ml2, mdist2);
      const u32x4 fv2 = v4_far_issue(out, nf2, md2, mdist2);
      v4_far_finish(L.win, out, R, nf, fv, md, mdist);
      const bool near = valid_m && md - mdist + ml > R;
      const unsigned long long nmask = __ballot(near);
      if (near) mlist[n_near + __builtin_amdgcn_mbcnt_hi((uint32_t)(nmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nmask, 0u))] = m;
      n_near += (uint32_t)__popcll(nmask);
      if (dbg) { dbg_matches += nmb; dbg_near += (uint32_t)__popcll(nmask); }
      m = m2; md = md2; ml = ml2; mdist = mdist2; nf = nf2; fv = fv2;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    RTOCK(3);
    if ((uint32_t)lane < n_near) m_next = mlist[lane];
    for (uint32_t k = 0; k < n_near; k += WAVE) {
      const uint32_t nmb = n_near - k < WAVE ? n_near - k : WAVE;
      const unsigned long long m = m_next;
      if (k + WAVE + (uint32_t)lane < n_near) m_next = mlist[k + WAVE + lane];
      const uint32_t md = (uint32_t)(m & 0xFFFFFFFFull), ml = (uint32_t)((m >> 32) & 0xFFFu), mdist = (uint32_t)(m >> 44);
      v4_near_batch(L.win, R, lane, (int)nmb, md, ml, mdist);
    }
    RTOCK(4);
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
    RTOCK(5);
  } else {
    for (uint32_t k = 0; k < tot_m; k += WAVE) {
      const uint32_t nmb = tot_m - k < WAVE ? tot_m - k : WAVE;
      const unsigned long long m = m_next;
      if (k + WAVE + (uint32_t)lane < tot_m) m_next = mlist[k + WAVE + lane];  // prefetch the next batch
      const uint32_t md = (uint32_t)(m & 0xFFFFFFFFull), ml = (uint32_t)((m >> 32) & 0xFFFu), mdist = (uint32_t)(m >> 44);
      v4_resolve_batch(out, lane, (int)nmb, md, ml, mdist);
    }
  }
}

#ifndef V4_WAVES_PER_EU
#define V4_WAVES_PER_EU 4
#endif
// DBG: the anatomy counters (BIOSCAN_DEBUG=1) are their own instantiation -- sixteen 64-bit cycle sums and seven event
// counters are ~45 of a wave's ~100 scalar registers, and carried as dead weight they spill (a third of the kernel's
// static vector instructions were v_readlane / v_writelane of spilled scalars).
template <int WPW, bool BOUNDED, bool DBG>
__global__ __launch_bounds__(WAVE * WPW, V4_WAVES_PER_EU) void k_bgzf_inflate_v4(const uint8_t* __restrict__ comp,
                                                           const uint64_t* __restrict__ blk_coff,
                                                           const uint64_t* __restrict__ blk_uoff, uint8_t* out_all,
                                                           uint32_t n_blocks, uint32_t* __restrict__ status,
                                                           uint32_t* counter, unsigned long long* scratch,
                                                           uint32_t scratch_stride, uint32_t* dbg_arg, uint32_t* slots, uint32_t n_slots,
                                                           uint32_t per_wave, const uint32_t* __restrict__ pre) {
  uint32_t* const dbg = DBG ? dbg_arg : nullptr;
  __shared__ V4Lds L_all[WPW];
  V4Lds& L = L_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
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
      // every resident wave holds at most one stride and there are more strides than resident waves, so a free one turns
      // up within a few probes; the probe count is bounded anyway: a wave that finds none takes no member and retires (the
      // members it would have taken are decoded by the waves that follow)
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
  uint32_t* ck = (uint32_t*)(mlist + V4_ML_ENTRIES);
  uint32_t dbg_rounds = 0, dbg_passes = 0, dbg_matches = 0, dbg_near = 0, dbg_minis = 0, dbg_idle = 0, dbg_hbm = 0;
  // expected length of the next DEFLATE block body: the previous block's; for the first member a wave takes, that member's
  // payload (a BGZF member is usually one block, and an overestimate only idles lanes behind the END-OF-BLOCK while an
  // underestimate costs whole rounds -- with the fixed 12 KiB guess the first member of a wave ran 2.4 extra rounds, which is
  // most of what a launch of a few thousand members costs: 5.4 rounds per member instead of 3)
  uint64_t pred_bits = 0;
#ifdef V4_FIXSTAT
  uint32_t fs_lanes[4] = {0, 0, 0, 0}, fs_iters[4] = {0, 0, 0, 0};  // lanes re-decoded by / runs of the 1st, 2nd, 3rd, later fix pass
#endif
  unsigned long long tc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tcx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t0 = 0;
#define TICK() (t0 = dbg ? clock64() : 0)
#define TOCK(i) do { if (dbg) { unsigned long long t1 = clock64(); tc[i] += t1 - t0; t0 = t1; } } while (0)
  // Compile-time experiment (tools/build_variant.sh): s_setprio around one phase, so that the SIMD's issue arbiter prefers the
  // waves that are in it.  V4_PRIO_DECODE / _WRITE / _RESOLVE = priority (1..3) of that phase, 0 elsewhere.
#ifdef V4_PRIO_DECODE
#define PRIO_DECODE(on) __builtin_amdgcn_s_setprio((on) ? V4_PRIO_DECODE : 0)
#else
#define PRIO_DECODE(on) do {} while (0)
#endif
#ifdef V4_PRIO_WRITE
#define PRIO_WRITE(on) __builtin_amdgcn_s_setprio((on) ? V4_PRIO_WRITE : 0)
#else
#define PRIO_WRITE(on) do {} while (0)
#endif
#ifdef V4_PRIO_RESOLVE
#define PRIO_RESOLVE(on) __builtin_amdgcn_s_setprio((on) ? V4_PRIO_RESOLVE : 0)
#else
#define PRIO_RESOLVE(on) do {} while (0)
#endif

  // the STOP tables (see the entry format) and K0's record pointer, once per wave
  {
    if (lane < 4) L.tab[V4_STOP_EOB + lane] = E4_PASS;
    if (lane == 0) { L.pre_lo = (uint32_t)(uintptr_t)pre; L.pre_hi = (uint32_t)((uintptr_t)pre >> 32); }
  }
  V4_SYNC();

  for (;;) {
    if constexpr (BOUNDED) {
      V4_SYNC();
      const uint32_t left = uni2(L.bnd_budget);
      if (left == 0u) break;
      V4_SYNC();
      if (lane == 0) L.bnd_budget = left - 1u;
    }
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
    // (the written-out loops address the tables from LDS address 0: true by construction, checked all the same)
    if (WPW == 1 && !V4_NO_ASM && uni2((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)L.tab) != 0u) {
      if (lane == 0) status[b] = INF_BAD_HEADER | (9u << 8);
      continue;
    }
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
          pr += (size_t)b * V4_PRE_DWORDS;
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
        V4_SYNC();
        ub_init(in, base32, skew + pre_bits, lane);
      } else if (btype == 1) {
        for (int i = lane; i < 320; i += WAVE) {
          uint8_t l;
          if (i < 144) l = 8; else if (i < 256) l = 9; else if (i < 280) l = 7; else if (i < 288) l = 8; else l = 5;
          L.b.lens[i] = l;
        }
        V4_SYNC();
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
          V4_SYNC();
        }
      }
      TOCK(5);
      P = ub_bitpos(in);   // the block's body starts here (the table builds below read the code lengths from LDS only)

      bool block_done = false;
      uint32_t force_dw = 0;  // != 0: the round is being re-run with short sub-streams (a pass ran out of checkpoint rows)
      const uint64_t block_P0 = P;
      // dwords per sub-stream of the round that starts at bit Pq of this block
      auto round_sub_dw = [&](uint64_t Pq) -> uint32_t {
        // A round should end with its block: it covers what is left of the predicted block length (the previous block's
        // length -- zlib and libdeflate cut blocks of similar size), with a little slack because an underestimate costs a
        // whole extra round and an overestimate only idle lanes behind the END-OF-BLOCK.
        const uint64_t rem_bits = end_bits > Pq ? end_bits - Pq : 0;
        const uint64_t used = Pq - block_P0;
        uint64_t want = pred_bits > used + pred_bits / 8 ? pred_bits - used : pred_bits / 8;
        want += want / 16 + 64;
        if (want > rem_bits) want = rem_bits;
#ifndef V4_NO_EXACT_FINAL
        // A member's FINAL block ends with its payload (at most 7 bits of padding behind the END-OF-BLOCK), so its length needs
        // no prediction: the rounds that are left share the remaining bits evenly and the last one ends with the block (a round
        // is a fraction of a block here -- 64 x V4_SUB_DW dwords -- so this sizes every round but a member's last few bits).
        if (uni2(L.blk_final)) {
          const uint32_t round_max = 64u * 32u * (uint32_t)V4_MAX_SUB_DW;   // (a payload is < 2^19 bits)
          const uint32_t rb = (uint32_t)rem_bits;
          const uint32_t rounds_left = (rb + round_max - 1u) / round_max;
          want = rounds_left > 1u ? (rb + rounds_left - 1u) / rounds_left : rb;
        }
#endif
        uint32_t sdw = (uint32_t)((want + 64ull * 32 - 1) / (64ull * 32));
        if (sdw > (uint32_t)V4_MAX_SUB_DW) sdw = V4_MAX_SUB_DW;
        if (sdw < 5) sdw = 5;
        return sdw;
      };
      // the NEXT round's dwords, asked for as soon as this round's decode passes have fixed where it ends and held in registers
      // while the round writes and resolves: staging them then costs three LDS stores instead of an HBM round trip (the stage
      // phase goes from 3.1 % to 1.4 % of the wave cycles; the launch gains 0.6 % -- other waves were hiding most of that wait)
      constexpr bool PF_ON = WPW == 1 && V4_PREFETCH != 0;   // (the shared-table shape of long members has no registers to spare)
      constexpr int V4_PF = (V4_STAGE_DW / 4 + WAVE - 1) / WAVE;
      u32x4 pf[V4_PF];
      uint64_t pf_wb = ~0ull;   // dword index the registers were loaded from (~0: nothing held)
      if constexpr (PF_ON) {
        // the first round's dwords travel while the tables are built
        const uint32_t n16n = (round_sub_dw(P) * 64u + V4_STAGE_SLACK) / 4u;
        const uint64_t wbn = P >> 5;
#pragma unroll
        for (int q = 0; q < V4_PF; q++) {
          const uint32_t i = (uint32_t)lane + (uint32_t)q * WAVE;
          if (i < n16n) pf[q] = ld16((const uint8_t*)(base32 + wbn) + 16u * i);
        }
        pf_wb = wbn;
      }
      {
        // literal/length sub-tables from the pool's high end, distance sub-tables from its low end; codes that need more
        // than the pool holds go to the wide-table kernel
        uint32_t lit_tot, dist_tot;
        int rc = v4_build(L, L.b.lens, 288, V4_LIT_ROOT, V4_LIT_BITS, V4_POOL_LO, V4_LIT_ROOT, L.b.lit_sorted, false, lane, lit_tot);
        if (rc) { st = rc == 2 ? (uint32_t)INF_RETRY : INF_BAD_CODE | (3u << 8); break; }
        rc = v4_build(L, L.b.lens + 288, 32, V4_DIST_ROOT, V4_DIST_BITS, V4_POOL_LO, V4_LIT_ROOT - lit_tot, L.b.dist_sorted, true, lane, dist_tot);
        if (rc) { st = rc == 2 ? (uint32_t)INF_RETRY : INF_BAD_CODE | (4u << 8); break; }
        if (lane == 0) L.lit_lo = V4_LIT_ROOT - lit_tot;
        V4_SYNC();
      }
      TOCK(0);

      // ---- rounds over the block body ----
      while (!block_done) {
        uint32_t sub_dw = round_sub_dw(P);
        if (force_dw) sub_dw = force_dw;
        const uint32_t subb = sub_dw * 32;
        uint32_t ovb = (subb * (uint32_t)V4_OV_QUARTERS) >> 2;   // pre-roll of the speculative lanes
        if (ovb < (uint32_t)V4_OV_MIN) ovb = V4_OV_MIN;
        if (ovb > (uint32_t)V4_OV_MAX) ovb = V4_OV_MAX;
        const uint64_t wb = P >> 5;
        v4_gsrc_t gsrc = (v4_gsrc_t)(base32 + wb);
        // stage the round's dwords: coalesced 16-byte loads (the payload is 4-byte aligned only; gfx950 takes unaligned vector
        // loads), the one HBM round trip of the round -- the decode loops below never touch global memory
        {
          const uint32_t n16 = (sub_dw * 64u + V4_STAGE_SLACK) / 4u;
          if (PF_ON && pf_wb == wb && !force_dw) {
#pragma unroll
            for (int q = 0; q < V4_PF; q++) {
              const uint32_t i = (uint32_t)lane + (uint32_t)q * WAVE;
              if (i < n16) { u32x4_raw r; r.x = pf[q].x; r.y = pf[q].y; r.z = pf[q].z; r.w = pf[q].w; *(u32x4_raw*)(L.stage + 4u * i) = r; }
            }
          } else
          for (uint32_t i = (uint32_t)lane; i < n16; i += WAVE) {
            const u32x4 v = ld16((const uint8_t*)(base32 + wb) + 16u * i);
            u32x4_raw r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
            *(u32x4_raw*)(L.stage + 4u * i) = r;
          }
          pf_wb = ~0ull;
          V4_SYNC();
        }
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
          // first symbol start at / after the lane's boundary (speculative)
          if constexpr (WPW == 1 && !V4_NO_ASM) start = v4_sync_asm(L, bnd - ov, bnd); else start = v4_sync(L, bnd - ov, bnd, limit, gsrc);
          TOCK(6);
          if (start >= limit) { end = start; }               // (a symbol that spans the whole sub-stream: the lane owns nothing)
          if constexpr (WPW == 1 && !V4_NO_ASM) ovf = v4_count_asm(L, start < limit, start, limit, ck, lane, end, acc, flags, cn);
          else ovf = v4_count(L, start < limit, start, limit, gsrc, ck, lane, end, acc, flags, cn);
        }
        dbg_passes++;
        for (int it = 0; it < 66 && !ovf; it++) {
          const unsigned long long stopm = __ballot(flags != 0);
          const int first_stop = stopm ? __builtin_ctzll(stopm) : 64;
          uint32_t pe = __shfl_up(end, 1, WAVE);
          const bool alive = lane > 0 && lane <= first_stop;
          const bool changed = alive && pe != start;
          if (__ballot(changed) == 0ull) break;
#ifdef V4_FIXSTAT
          { const int k = it < 3 ? it : 3; fs_lanes[k] += (uint32_t)__popcll(__ballot(changed)); fs_iters[k]++; }
#endif
          if (changed) start = pe;
          // a lane whose corrected start already lies beyond its limit owns no symbols
          if (changed && start >= limit) { end = start; acc = 0; flags = 0; cn = 0; }
          if constexpr (WPW == 1 && !V4_NO_ASM) ovf = v4_count_asm(L, changed && start < limit, start, limit, ck, lane, end, acc, flags, cn);
          else ovf = v4_count(L, changed && start < limit, start, limit, gsrc, ck, lane, end, acc, flags, cn, 1);
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
        if (PF_ON && !stopm) {
          // the block goes on: its next round starts at the end of the last lane's last symbol -- the same loads, with the same
          // bounds, that the staging above would issue then
          const uint64_t Pn = (wb << 5) + __builtin_amdgcn_readlane(end, last);
          if (Pn < end_bits + 64) {
            const uint32_t n16n = (round_sub_dw(Pn) * 64u + V4_STAGE_SLACK) / 4u;
            const uint64_t wbn = Pn >> 5;
#pragma unroll
            for (int q = 0; q < V4_PF; q++) {
              const uint32_t i = (uint32_t)lane + (uint32_t)q * WAVE;
              if (i < n16n) pf[q] = ld16((const uint8_t*)(base32 + wbn) + 16u * i);
            }
            pf_wb = wbn;
          }
        }
        // ---- write phase: the round's segments, in output order, 64 per mini-round ----
        const uint32_t nseg = (valid && start < limit) ? 1u + cn : 0u;
        uint32_t n_seg_tot;
        const uint32_t segbase = wave_excl_scan_u32(nseg, lane, &n_seg_tot);
        const uint32_t segend = segbase + nseg;
        uint32_t n_take = 0;  // segments of the current mini-round: 64, or as many as fit in the LDS window
        // A mini-round's lanes read their segment's checkpoint rows from the wave's scratch: one L2 round trip.  V4_PIPE_CK=1 asks
        // for the rows as soon as the PREVIOUS mini-round knows how many segments it takes, so that its write and resolve phases
        // hide the trip.  Measured (262144 members, same box): 45.28 ms against 45.14 ms without -- the kernel is VALU-bound at
        // four waves per SIMD and another wave already runs while this one waits; the seven registers the rows in flight take
        // spill two.  Kept as a switch, off.
        uint32_t seg_n = 0;   // the segment the rows in flight belong to: owner lane | row k << 6 | "there is one" << 31
        uint32_t l_p0 = 0, l_a0 = 0, l_st0 = 0, l_p1 = 0, l_a1 = 0;
        auto seg_issue = [&](uint32_t s0) {
          const uint32_t g = s0 + (uint32_t)lane;
          const bool has_n = g < n_seg_tot;
          // owner of segment g: the first lane whose segments end after g (segend is non-decreasing)
          int lo = 0, hi = WAVE;
#pragma unroll
          for (int step = 0; step < 7; step++) {  // 65 possible answers
            const int mid = (lo + hi) >> 1;
            const uint32_t v = (uint32_t)__shfl((int)segend, mid & 63, WAVE);
            if (lo < hi) { if (v <= g) lo = mid + 1; else hi = mid; }
          }
          const int own_n = lo < WAVE ? lo : WAVE - 1;
          const uint32_t own_base = (uint32_t)__shfl((int)segbase, own_n, WAVE);   // (every lane takes part: a lane may own segments of a mini-round it has none in)
          const uint32_t k_n = has_n ? g - own_base : 0u;
          seg_n = (uint32_t)own_n | (k_n << 6) | (has_n ? 0x80000000u : 0u);
          const uint32_t o_cn = (uint32_t)__shfl((int)cn, own_n, WAVE);
          if (has_n && k_n > 0) {
            const uint32_t* q = ck + k_n * V4_CK_ROW + (uint32_t)own_n;        // row k = state after k * V4_CK_STEPS steps
            l_p0 = q[0]; l_a0 = q[64]; l_st0 = q[128];
          }
          if (has_n && k_n < o_cn) {
            const uint32_t* q = ck + (k_n + 1u) * V4_CK_ROW + (uint32_t)own_n;
            l_p1 = q[0]; l_a1 = q[64];
          }
        };
        if (V4_PIPE_CK && n_seg_tot) seg_issue(0);
        for (uint32_t s0 = 0; s0 < n_seg_tot; s0 += n_take) {
          if (!V4_PIPE_CK) seg_issue(s0);
          bool has = (seg_n >> 31) != 0u;
          const int own = (int)(seg_n & 63u);
          const uint32_t k = (seg_n >> 6) & 0x1FFFFFFu;
          const uint32_t o_start = (uint32_t)__shfl((int)start, own, WAVE);
          const uint32_t o_obase = (uint32_t)__shfl((int)obase, own, WAVE);
          const uint32_t o_mbase = (uint32_t)__shfl((int)mbase, own, WAVE);
          const uint32_t o_cn = (uint32_t)__shfl((int)cn, own, WAVE);
          const uint32_t o_acc = (uint32_t)__shfl((int)acc, own, WAVE);
          const uint32_t o_limit = rel0 + (uint32_t)(own + 1) * subb;
          uint32_t p0 = o_start, a0 = 0, st0 = V4_LIT_ROOT | ((uint32_t)V4_LIT_BITS << 12), p1 = 0xFFFFFFFFu, a1 = o_acc;
          if (has && k > 0) { p0 = l_p0; a0 = l_a0; st0 = l_st0; }
          if (has && k < o_cn) { p1 = l_p1; a1 = l_a1; }
          const uint32_t seg_out = has ? (a1 & 0xFFFFFu) - (a0 & 0xFFFFFu) : 0u;
          const uint32_t seg_m = has ? (a1 >> 20) - (a0 >> 20) : 0u;
          const uint32_t my_opos = o_obase + (a0 & 0xFFFFFu);
          const uint32_t my_mabs = o_mbase + (a0 >> 20);
          const uint32_t R = __builtin_amdgcn_readlane(my_opos, 0);  // first output byte of the mini-round
          const uint32_t M0 = __builtin_amdgcn_readlane(my_mabs, 0);
          // Segments are consecutive in the output, so those whose bytes end inside the LDS window are a prefix of the
          // lanes: the mini-round takes that prefix (normally all 64) and the next one starts behind it.  A single segment
          // larger than the window (forty 258-byte matches) goes through HBM on its own.
          const uint32_t n_fit = (uint32_t)__popcll(__ballot(has && my_opos + seg_out - R <= (uint32_t)V4_WIN && my_mabs + seg_m - M0 <= V4_LCAP));
          const bool use_win = n_fit != 0u;
          n_take = use_win ? n_fit : 1u;
          has = has && (uint32_t)lane < n_take;
          const uint32_t nl = n_take - 1u;  // last lane with a segment
          const uint32_t out_s = __builtin_amdgcn_readlane(my_opos + seg_out, nl) - R;
          const uint32_t m_s = __builtin_amdgcn_readlane(my_mabs + seg_m, nl) - M0;
          if (!use_win && m_s > (uint32_t)V4_ML_ENTRIES) { st = INF_OVERRUN | (1u << 8); break; }
          if (dbg && !use_win) dbg_hbm++;
          if (V4_PIPE_CK && s0 + n_take < n_seg_tot) seg_issue(s0 + n_take);
          {
            uint32_t f2 = 0;
            const uint32_t tb0 = st0 & 0xFFFu, mb0 = (st0 >> 12) & 15u, ml0 = st0 >> 16;
#ifndef V4_ABLATE_WRITE
            if (dbg) { const unsigned long long t1 = clock64(); tcx[0] += t1 - t0; }
            const unsigned long long tw0 = dbg ? clock64() : 0;
            PRIO_WRITE(1);
            if (use_win) {
              if constexpr (WPW == 1 && !V4_NO_ASM) f2 = v4_write_win_asm(L, has, p0, tb0, mb0, ml0, o_limit, p1, my_opos - R, R, (my_mabs - M0) * 2u);
              else f2 = v4_write<2>(L, has, p0, tb0, mb0, ml0, o_limit, p1, out, my_opos, mlist, my_mabs - M0, R, gsrc);
            }
#ifdef V4_GUARD
            else f2 = v4_write<1>(L, has, p0, tb0, mb0, ml0, o_limit, p1, out, my_opos, mlist, my_mabs - M0, isize, gsrc);
#else
            else f2 = v4_write<1>(L, has, p0, tb0, mb0, ml0, o_limit, p1, out, my_opos, mlist, my_mabs - M0, R, gsrc);
#endif
            PRIO_WRITE(0);
            if (dbg) tcx[1] += clock64() - tw0;
#endif
            dbg_minis++;
            if (__ballot(f2 & F_BAD) != 0ull) { st = INF_BAD_DIST; break; }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          TOCK(3);
#ifndef V4_ABLATE_RESOLVE
          PRIO_RESOLVE(1);
          v4_resolve(L, out, mlist, lane, R, out_s, m_s, use_win, dbg != nullptr, dbg_matches, dbg_near, tcx);
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
#if defined(V4_ABLATE_WRITE) || defined(V4_ABLATE_RESOLVE)
    st = INF_OK;  // timing-only build: the bytes are wrong on purpose
#endif
    if (lane == 0) {
      status[b] = st;
      if (st == INF_RETRY) atomicAdd(counter + 1, 1u);   // the launch that follows (inflate_v3.hip, retry mode) looks here first
    }
  }
  if constexpr (BOUNDED) {
    V4_SYNC();
    if (lane == 0 && L.bnd_slot != 0xFFFFFFFFu) atomicExch(&slots[L.bnd_slot], 0u);  // (the scratch carries nothing from one owner to the next)
  }
  if (dbg && lane == 0) {
    atomicAdd(&dbg[0], dbg_rounds);
    atomicAdd(&dbg[1], dbg_passes);
    for (int i = 0; i < 5; i++) atomicAdd((unsigned long long*)(dbg + 2) + i, tc[i]);
    atomicAdd((unsigned long long*)(dbg + 26), tc[5]);
    atomicAdd((unsigned long long*)(dbg + 28), tc[6]);
    for (int i = 0; i < 8; i++) atomicAdd((unsigned long long*)(dbg + 32) + i, tcx[i]);
    atomicAdd(&dbg[12], dbg_matches);
    atomicAdd(&dbg[13], dbg_near);
    atomicAdd(&dbg[22], dbg_minis);
    atomicAdd(&dbg[23], dbg_idle);
    atomicAdd(&dbg[24], dbg_hbm);
#ifdef V4_FIXSTAT
    for (int k = 0; k < 4; k++) { atomicAdd(&dbg[14 + k], fs_lanes[k]); atomicAdd(&dbg[18 + k], fs_iters[k]); }
#endif
  }
}

void v4_guard_report() {
#ifdef V4_GUARD
  unsigned int w[8] = {0};
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(w, HIP_SYMBOL(v4_guard_word), sizeof w);
  fprintf(stderr, "[v2 guard] mask=%#x src_idx=%u lit_opos=%u mpos=%u resolve=%u resolve_win=%u\n", w[0], w[1], w[2], w[3], w[4], w[5]);
#endif
}

int v4_resident_wg_per_cu() {
  int n = 0;
  // resident WAVES per CU (= members decoded concurrently per CU)
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_bgzf_inflate_v4<V4_WAVES_PER_WG, false, false>, WAVE * V4_WAVES_PER_WG, 0) != hipSuccess || n < 1) n = 8;
  return n * V4_WAVES_PER_WG;
}
void launch_bgzf_inflate_v4(const uint8_t* comp, const uint64_t* blk_coff, const uint64_t* blk_uoff, uint8_t* out,
                            uint32_t n_blocks, uint32_t* status, uint32_t* counter, unsigned long long* scratch,
                            uint32_t grid, uint32_t* dbg, hipStream_t st, uint32_t* slots, uint32_t n_slots, uint32_t per_wave, uint32_t wpw,
                            const uint32_t* pre) {
  if (!n_blocks) return;
  if (wpw != 1) wpw = V4_BOUNDED_WPW;
  (void)hipMemsetAsync(counter, 0, 4, st);
  if (slots) {
    // bounded: workgroups of `wpw` waves, `per_wave` members each; scratch strides handed out through `slots`
    if (!per_wave) per_wave = 1;
    const uint32_t per_wg = wpw * per_wave;
    const uint32_t g = (n_blocks + per_wg - 1) / per_wg;
    if (wpw == 1)
      hipLaunchKernelGGL((k_bgzf_inflate_v4<1, true, false>), dim3(g), dim3(WAVE), 0, st, comp, blk_coff, blk_uoff, out, n_blocks,
                         status, counter, scratch, (uint32_t)V4_SCRATCH_STRIDE, dbg, slots, n_slots, per_wave, pre);
    else
      hipLaunchKernelGGL((k_bgzf_inflate_v4<V4_BOUNDED_WPW, true, false>), dim3(g), dim3(WAVE * V4_BOUNDED_WPW), 0, st, comp, blk_coff, blk_uoff, out, n_blocks,
                         status, counter, scratch, (uint32_t)V4_SCRATCH_STRIDE, dbg, slots, n_slots, per_wave, pre);
  } else {
    uint32_t g = grid < n_blocks ? grid : n_blocks;
    g = (g + V4_WAVES_PER_WG - 1) / V4_WAVES_PER_WG;  // `grid` counts waves; the scratch holds grid + V4_WAVES_PER_WG strides
    if (dbg)
      hipLaunchKernelGGL((k_bgzf_inflate_v4<V4_WAVES_PER_WG, false, true>), dim3(g), dim3(WAVE * V4_WAVES_PER_WG), 0, st, comp, blk_coff, blk_uoff, out, n_blocks,
                         status, counter, scratch, (uint32_t)V4_SCRATCH_STRIDE, dbg, nullptr, 0u, 0u, pre);
    else
      hipLaunchKernelGGL((k_bgzf_inflate_v4<V4_WAVES_PER_WG, false, false>), dim3(g), dim3(WAVE * V4_WAVES_PER_WG), 0, st, comp, blk_coff, blk_uoff, out, n_blocks,
                         status, counter, scratch, (uint32_t)V4_SCRATCH_STRIDE, nullptr, nullptr, 0u, 0u, pre);
  }
#ifdef V4_GUARD
  v4_guard_report();
#endif
#ifdef V4_UTIL
  {
    unsigned long long u[4] = {0, 0, 0, 0};
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(u, HIP_SYMBOL(v4_util), sizeof u);
    if (n_blocks > 1000)
      fprintf(stderr, "[v3 util] first count pass: %llu steps, %.1f lanes busy per step; fix passes: %llu steps, %.1f lanes busy per step (cumulative)\n",
              u[0], u[0] ? (double)u[1] / u[0] : 0.0, u[2], u[2] ? (double)u[3] / u[2] : 0.0);
  }
#endif
}

}  // namespace bioscan
