// fastq_kernels.hip -- FASTQ record framing + field extract -> Arrow scatter (gfx950).
//
// Replaces the per-record loop of the reference's FastqExec
// (bio-format-fastq/src/physical_exec.rs:393-465 `batch_producer`, :184-248 resync) and
// noodles-fastq 0.23.0 `Reader::read_record` (un-vendored).  Byte work only.
// Stages: resync (k_fastq_sync) -> newline index of the partition's text (k_nl_count / k_nl_write) -> records to the four
// Utf8 columns in two passes without per-row arrays (k_fastq_pass1 / k_fastq_pass2, the scheme of bam_rows.hip).
#include "kernels.h"
#include <stdlib.h>

namespace bioscan {

#define WAVE 64

// ---- wave-parallel byte search: first position in [from, to) holding `byte`, or ~0 ---------------
__device__ __forceinline__ uint64_t wave_find(const uint8_t* u, uint64_t from, uint64_t to, uint8_t byte, int lane) {
  for (uint64_t p0 = from; p0 < to; p0 += WAVE) {
    const uint64_t p = p0 + lane;
    const bool hit = p < to && u[p] == byte;
    const unsigned long long m = __ballot(hit);
    if (m) return p0 + (uint64_t)__builtin_ctzll(m);
  }
  return ~0ull;
}

// Resync (physical_exec.rs:184-248): walk buffered windows from `start`; inside one window look for
// the first '@' whose line+2 starts with '+'.  win_end[j] = end of window j; win_coff[j] / win_next[j]
// = compressed offset the reader reports inside window j / once window j is exhausted (BGZF only;
// check_end = 0 for plain files, whose resync has no end test).  result[0] = position, or ~0 when the
// stream ended.
__global__ __launch_bounds__(WAVE) void k_fastq_sync(const uint8_t* __restrict__ u, uint64_t start, uint64_t ulen,
                                                      const uint64_t* __restrict__ win_end,
                                                      const uint64_t* __restrict__ win_coff,
                                                      const uint64_t* __restrict__ win_next, uint32_t n_win,
                                                      uint64_t end_comp, int check_end, unsigned long long* result) {
  const int lane = threadIdx.x;
  uint64_t x = start;
  uint32_t j = 0;
  while (j < n_win && win_end[j] <= x) j++;  // window containing x (x is a window start after a seek)
  uint64_t res = ~0ull;
  for (;;) {
    if (j >= n_win || x >= ulen) { res = x < ulen ? x : ~0ull; break; }
    if (check_end) {
      const uint64_t vc = x < win_end[j] ? win_coff[j] : win_next[j];
      if (vc >= end_comp) { res = x; break; }
    }
    if (x >= win_end[j]) { j++; continue; }  // fill_buf loads the next window
    const uint64_t wend = win_end[j];
    const uint64_t at = wave_find(u, x, wend, '@', lane);
    if (at == ~0ull) { x = wend; continue; }
    const uint64_t l1 = wave_find(u, at, wend, '\n', lane);
    if (l1 != ~0ull) {
      const uint64_t l2 = wave_find(u, l1 + 1, wend, '\n', lane);
      if (l2 != ~0ull && l2 + 1 < wend && u[l2 + 1] == '+') { res = at; break; }
      x = l1 + 1;
    } else {
      x = wend;
    }
  }
  if (lane == 0) result[0] = res;
}
void launch_fastq_sync(const uint8_t* u, uint64_t start, uint64_t ulen, const uint64_t* win_end, const uint64_t* win_coff,
                       const uint64_t* win_next, uint32_t n_win, uint64_t end_comp, int check_end, unsigned long long* result,
                       hipStream_t st) {
  hipLaunchKernelGGL(k_fastq_sync, dim3(1), dim3(WAVE), 0, st, u, start, ulen, win_end, win_coff, win_next, n_win, end_comp,
                     check_end, result);
}

// ---- newline index ------------------------------------------------------------------------------------
// Tile = 16 KiB per 256-thread workgroup, 64 contiguous bytes per thread (adjacent lanes read adjacent
// 64-byte lines).  '\n' bytes are found 8 at a time with an exact SWAR zero-byte test; pass 1 counts
// per tile, a scan gives tile bases, pass 2 recounts and writes the positions in order.
// newline index entry: position [0:47] | first byte of the next line [48:55] | bit 63: the byte before is '\r'
constexpr uint64_t NL_POS = (1ull << 48) - 1, NL_CR = 1ull << 63;
constexpr int NL_NEXT_SHIFT = 48;
constexpr int NL_CHUNK = 16384;
constexpr int NL_PER_THREAD = 64;
struct __attribute__((packed, aligned(1))) nl_u64 { uint64_t v; };
struct __attribute__((packed, aligned(1))) nl_u64x2 { uint64_t a, b; };

// bit 8k+7 of the result is set iff byte k of w equals '\n'
__device__ __forceinline__ uint64_t nl_mask8(uint64_t w) {
  const uint64_t x = w ^ 0x0A0A0A0A0A0A0A0Aull;
  const uint64_t t = (x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full;
  return ~(t | x | 0x7F7F7F7F7F7F7F7Full);
}
// masks of the 8 words of this thread's 64 bytes [a, a+64) clipped to hi (bytes past hi never match)
__device__ __forceinline__ uint32_t nl_thread_masks(const uint8_t* __restrict__ u, uint64_t a, uint64_t hi, uint64_t m[8],
                                                     uint64_t* words = nullptr) {
  uint32_t n = 0;
  uint64_t wv[8];
  if (a + 64 <= hi) {  // whole 64 bytes inside the range: four 16-byte loads
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const nl_u64x2 v = *(const nl_u64x2*)(u + a + 16 * k);
      wv[2 * k] = v.a; wv[2 * k + 1] = v.b;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint64_t p = a + 8 * k;
      uint64_t w = 0;
      if (p + 8 <= hi) w = ((const nl_u64*)(u + p))->v;
      else if (p < hi) { for (uint64_t q = p; q < hi; q++) w |= (uint64_t)u[q] << (8 * (q - p)); }
      wv[k] = w;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint64_t p = a + 8 * k;
    m[k] = p < hi ? nl_mask8(wv[k]) : 0ull;
    if (words) words[k] = wv[k];
    n += (uint32_t)__popcll(m[k]);
  }
  return n;
}
// add: the counts are added to cnt (tiles K2 has counted the member bytes of -- crc32.hip -- get the bytes in front of the first
// member, the record carried from the previous chunk, this way)
__global__ __launch_bounds__(256) void k_nl_count(const uint8_t* __restrict__ u, uint64_t lo, uint64_t hi, uint32_t* __restrict__ cnt, int add) {
  __shared__ uint32_t s_w[4];
  const uint64_t a = lo + (uint64_t)blockIdx.x * NL_CHUNK + (uint64_t)threadIdx.x * NL_PER_THREAD;
  uint64_t m[8];
  uint32_t n = a < hi ? nl_thread_masks(u, a, hi, m) : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) n += __shfl_down(n, d, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (add) { if (t) atomicAdd(&cnt[blockIdx.x], t); } else cnt[blockIdx.x] = t;
  }
}
// number of entries of the newline index whose position is below x (entries are in position order)
__global__ void k_nl_lower_bound(const uint64_t* __restrict__ nl, uint64_t n, uint64_t x, unsigned long long* out) {
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if ((nl[mid] & NL_POS) < x) lo = mid + 1; else hi = mid;
  }
  *out = lo;
}
void launch_nl_lower_bound(const uint64_t* nl, uint64_t n, uint64_t x, unsigned long long* out, hipStream_t st) {
  hipLaunchKernelGGL(k_nl_lower_bound, dim3(1), dim3(1), 0, st, nl, n, x, out);
}
__global__ __launch_bounds__(256) void k_nl_write(const uint8_t* __restrict__ u, uint64_t lo, uint64_t hi,
                                                   const uint64_t* __restrict__ base, uint64_t* __restrict__ nl) {
  __shared__ uint32_t s_w[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t a = lo + (uint64_t)blockIdx.x * NL_CHUNK + (uint64_t)threadIdx.x * NL_PER_THREAD;
  uint64_t m[8], w[8];
  const uint32_t n = a < hi ? nl_thread_masks(u, a, hi, m, w) : 0u;
  uint32_t inc = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  uint32_t wbase = 0;
  for (int i = 0; i < wv; i++) wbase += s_w[i];
  uint64_t o = base[blockIdx.x] + wbase + (inc - n);
  if (n) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      uint64_t mk = m[k];
      while (mk) {
        const int bit = __builtin_ctzll(mk);
        const int b = bit >> 3;
        const uint64_t p = a + 8 * k + (uint64_t)b;
        // the bytes around the newline ride along, so that the field kernel does not have to touch the text again for
        // CRLF and '@' / '+' checks: they come from this thread's own words except at the ends of its 64 bytes
        uint32_t pb, nb;
        if (b > 0) pb = (uint32_t)(w[k] >> (8 * (b - 1))) & 0xFFu;
        else if (k > 0) pb = (uint32_t)(w[k - 1] >> 56);
        else pb = p > lo ? u[p - 1] : 0u;
        if (b < 7) nb = (uint32_t)(w[k] >> (8 * (b + 1))) & 0xFFu;
        else if (k < 7) nb = (uint32_t)w[k + 1] & 0xFFu;
        else nb = p + 1 < hi ? u[p + 1] : 0u;
        if (p + 1 >= hi) nb = 0u;
        nl[o++] = p | (pb == '\r' ? NL_CR : 0ull) | ((uint64_t)nb << NL_NEXT_SHIFT);
        mk &= mk - 1;
      }
    }
  }
}
uint64_t nl_chunks(uint64_t lo, uint64_t hi) { return hi > lo ? (hi - lo + NL_CHUNK - 1) / NL_CHUNK : 0; }
void launch_nl_count(const uint8_t* u, uint64_t lo, uint64_t hi, uint32_t* cnt, hipStream_t st, bool add) {
  const uint64_t n = nl_chunks(lo, hi);
  if (!n) return;
  hipLaunchKernelGGL(k_nl_count, dim3((uint32_t)n), dim3(256), 0, st, u, lo, hi, cnt, add ? 1 : 0);
}
void launch_nl_write(const uint8_t* u, uint64_t lo, uint64_t hi, const uint64_t* base, uint64_t* nl, hipStream_t st) {
  const uint64_t n = nl_chunks(lo, hi);
  if (!n) return;
  hipLaunchKernelGGL(k_nl_write, dim3((uint32_t)n), dim3(256), 0, st, u, lo, hi, base, nl);
}

// ---- record fields ------------------------------------------------------------------------------------
// Record r spans lines 4r..4r+3 counted from x0 (the first record start).  nl[] holds the newline
// positions >= x0 in order; a last line without '\n' ends at `eof` (only legal at the end of the data).
// Per record: source offset + length of name, description, sequence, quality; description
// validity (NULL when empty, physical_exec.rs:430-434); err: 1 = missing '@', 2 = missing '+'.
// bit 8k+7 set iff byte k of w equals c (exact SWAR zero-byte test)
__device__ __forceinline__ uint64_t eq_mask8(uint64_t w, uint64_t c8) {
  const uint64_t x = w ^ c8;
  const uint64_t t = (x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full;
  return ~(t | x | 0x7F7F7F7F7F7F7F7Full);
}
// ---- records -> Utf8 columns in two passes (declared in kernels.h) -------------------------------------------------
struct FqGeom { uint64_t off[4]; uint32_t len[4]; bool dvalid; uint32_t err; };
// the fields of record r (the logic of k_fastq_fields: lines 4r .. 4r + 3 from the newline index, CRLF, a last line without
// '\n', the name / description split eight bytes at a time)
__device__ __forceinline__ FqGeom fq_geom(const uint8_t* __restrict__ u, uint64_t x0, uint64_t eof, const uint64_t* __restrict__ nl,
                                          uint64_t n_nl, uint64_t r) {
  FqGeom g;
  uint64_t q[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const uint64_t li = 4 * r + k;  // entry li - 1
    q[k] = (li >= 1 && li - 1 < n_nl) ? nl[li - 1] : ~0ull;
  }
  uint64_t s[4], e[4];
  uint32_t first[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint64_t li = 4 * r + k;
    const bool have_prev = li >= 1 && li - 1 < n_nl;
    s[k] = li == 0 ? x0 : (have_prev ? (q[k] & NL_POS) + 1 : eof);
    first[k] = li == 0 ? (x0 < eof ? u[x0] : 0u) : (have_prev ? (uint32_t)(q[k] >> NL_NEXT_SHIFT) & 0xFFu : 0u);
    uint64_t en;
    if (li < n_nl) {
      en = q[k + 1] & NL_POS;
      if (en > s[k] && (q[k + 1] & NL_CR)) en--;  // CRLF
    } else {
      en = eof;                                  // a last line without '\n'
      if (en > s[k] && u[en - 1] == '\r') en--;
    }
    e[k] = en < s[k] ? s[k] : en;
  }
  g.err = 0;
  if (first[0] != '@') g.err = 1;
  else if (s[2] < eof && first[2] != '+') g.err = 2;
  const uint64_t d0 = s[0] + 1 < e[0] ? s[0] + 1 : e[0];
  uint64_t sp = e[0];
  for (uint64_t p = d0; p < e[0]; p += 8) {
    uint64_t w;
    if (p + 8 <= eof) w = ((const nl_u64*)(u + p))->v;
    else { w = 0; for (uint64_t t = p; t < eof; t++) w |= (uint64_t)u[t] << (8 * (t - p)); }
    uint64_t m = eq_mask8(w, 0x2020202020202020ull) | eq_mask8(w, 0x0909090909090909ull);
    const uint64_t left = e[0] - p;
    if (left < 8) m &= (1ull << (8 * left)) - 1ull;
    if (m) { sp = p + (uint64_t)(__builtin_ctzll(m) >> 3); break; }
  }
  g.off[0] = d0; g.len[0] = (uint32_t)(sp - d0);
  g.off[1] = sp < e[0] ? sp + 1 : e[0]; g.len[1] = (uint32_t)(e[0] - g.off[1]);
  g.dvalid = g.len[1] != 0;
  g.off[2] = s[1]; g.len[2] = (uint32_t)(e[1] - s[1]);
  g.off[3] = s[3]; g.len[3] = (uint32_t)(e[3] - s[3]);
  return g;
}
__device__ __forceinline__ uint64_t fq_wave_incl_scan(uint64_t v, int lane) {
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const uint32_t lo = __shfl_up((uint32_t)v, d, WAVE);
    const uint32_t hi = __shfl_up((uint32_t)(v >> 32), d, WAVE);
    if (lane >= d) v += ((uint64_t)hi << 32) | lo;
  }
  return v;
}
__device__ __forceinline__ uint64_t fq_tile_base(const uint64_t* __restrict__ tile_sums, int k, uint64_t tile, uint64_t n_tiles) {
  const uint64_t n_groups = (n_tiles + TS_GROUP - 1) / TS_GROUP;
  const uint64_t* aux = tile_sums + 6 * (n_tiles + 1);
  return tile_sums[(uint64_t)k * (n_tiles + 1) + tile] + aux[(uint64_t)k * (n_groups + 1) + tile / TS_GROUP];
}

__global__ __launch_bounds__(ROWS_TILE) void k_fastq_pass1(const uint8_t* __restrict__ u, uint64_t x0, uint64_t eof, const uint64_t* __restrict__ nl,
                                                           uint64_t n_nl, uint64_t n, FqCols c, uint64_t n_tiles,
                                                           uint64_t* __restrict__ tile_sums, uint32_t* err) {
  __shared__ uint64_t s_w[4][ROWS_TILE / WAVE];
  const uint64_t r = (uint64_t)blockIdx.x * ROWS_TILE + threadIdx.x;
  const bool act = r < n;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  FqGeom g{};
  if (act) {
    g = fq_geom(u, x0, eof, nl, n_nl, r);
    if (g.err) atomicExch(err, g.err);
  }
  if (c.v_desc) {
    const unsigned long long m = __ballot(act && g.dvalid);
    if (lane == 0 && act) c.v_desc[r >> 6] = m;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (!((c.want >> k) & 1u)) continue;
    uint64_t v = act ? g.len[k] : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint64_t)__shfl_down((uint32_t)v, d, WAVE) | ((uint64_t)__shfl_down((uint32_t)(v >> 32), d, WAVE) << 32);
    if (lane == 0) s_w[k][w] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4 && ((c.want >> threadIdx.x) & 1u)) {
    uint64_t t = 0;
    for (int q = 0; q < ROWS_TILE / WAVE; q++) t += s_w[threadIdx.x][q];
    tile_sums[(uint64_t)threadIdx.x * (n_tiles + 1) + blockIdx.x] = t;
  }
}

// first byte of every batch: base of the tile that holds the batch's first row + the rows of that tile in front of it
__global__ __launch_bounds__(ROWS_TILE) void k_fastq_batch_bases(const uint8_t* __restrict__ u, uint64_t x0, uint64_t eof, const uint64_t* __restrict__ nl,
                                                                 uint64_t n_nl, uint64_t n, FqCols c, uint32_t bs, uint32_t phase, uint64_t n_tiles,
                                                                 const uint64_t* __restrict__ tile_sums) {
  __shared__ uint64_t s_w[4][ROWS_TILE / WAVE];
  const uint64_t b = blockIdx.x;
  // (`phase` rows of the chunk's first batch were delivered by the previous chunk: batch 0 holds bs - phase rows)
  const uint64_t s = b ? b * (uint64_t)bs - phase : 0;
  const uint64_t tile = s / ROWS_TILE, part = s % ROWS_TILE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint64_t r = tile * ROWS_TILE + threadIdx.x;
  FqGeom g{};
  if (threadIdx.x < part && r < n) g = fq_geom(u, x0, eof, nl, n_nl, r);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (!((c.want >> k) & 1u)) continue;
    uint64_t v = g.len[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += (uint64_t)__shfl_down((uint32_t)v, d, WAVE) | ((uint64_t)__shfl_down((uint32_t)(v >> 32), d, WAVE) << 32);
    if (lane == 0) s_w[k][w] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4 && ((c.want >> threadIdx.x) & 1u)) {
    uint64_t t = tile < n_tiles ? fq_tile_base(tile_sums, (int)threadIdx.x, tile, n_tiles) : tile_sums[(uint64_t)threadIdx.x * (n_tiles + 1) + n_tiles];
    for (int q = 0; q < ROWS_TILE / WAVE; q++) t += s_w[threadIdx.x][q];
    c.base[threadIdx.x][b] = t;
  }
}

struct __attribute__((packed, aligned(1))) fq_u32x4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) fq_u64 { uint64_t v; };
struct __attribute__((packed, aligned(1))) fq_u32 { uint32_t v; };
struct __attribute__((packed, aligned(1))) fq_u16 { uint16_t v; };
#ifndef FQ_UNROLL
#define FQ_UNROLL 2
#endif
__global__ __launch_bounds__(ROWS_TILE) void k_fastq_pass2(const uint8_t* __restrict__ u, uint64_t x0, uint64_t eof, const uint64_t* __restrict__ nl,
                                                           uint64_t n_nl, uint64_t n, FqCols c, uint32_t bs, uint32_t phase, uint64_t n_tiles,
                                                           const uint64_t* __restrict__ tile_sums) {
  __shared__ uint64_t s_w[4][ROWS_TILE / WAVE];
  __shared__ uint64_t s_src[4][ROWS_TILE], s_dst[4][ROWS_TILE];
  __shared__ uint32_t s_len[4][ROWS_TILE];
  __shared__ uint32_t s_cs[ROWS_TILE / WAVE][WAVE];   // per wave: first item number of every row (group phase)
  const uint64_t tile = blockIdx.x;
  const uint64_t r = tile * ROWS_TILE + threadIdx.x;
  const bool act = r < n;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  FqGeom g{};
  if (act) g = fq_geom(u, x0, eof, nl, n_nl, r);
  uint64_t off[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    off[k] = 0;
    if (!((c.want >> k) & 1u)) continue;
    const uint64_t inc = fq_wave_incl_scan((uint64_t)g.len[k], lane);
    if (lane == 63) s_w[k][w] = inc;
    off[k] = inc - g.len[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (!((c.want >> k) & 1u)) continue;
    uint64_t base = fq_tile_base(tile_sums, k, tile, n_tiles);
    for (int q = 0; q < w; q++) base += s_w[k][q];
    off[k] += base;
  }
  if (act) {
    // per-batch int32 offsets: entry j of batch b, and the closing entry when this is the batch's (or the scan's) last row
    const uint64_t v = r + phase;
    const uint64_t b = v / bs, j = b ? v - b * (uint64_t)bs : r;
    const bool closes = (v + 1) % bs == 0 || r + 1 == n;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (!((c.want >> k) & 1u)) continue;
      const uint64_t bb = c.base[k][b];
      int32_t* o32 = c.off32[k] + b * ((uint64_t)bs + 1);
      o32[j] = (int32_t)(off[k] - bb);
      if (closes) o32[j + 1] = (int32_t)(off[k] + g.len[k] - bb);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) { s_src[k][threadIdx.x] = g.off[k]; s_dst[k][threadIdx.x] = off[k]; s_len[k][threadIdx.x] = act ? g.len[k] : 0u; }
  __syncthreads();
  // The bytes.  Measured: with one row per 16-lane group (a 101-byte read fills 7 lanes of 16, a 23-byte description 2, a name
  // 1) this phase ran at half of what the same phase of bam_rows.hip reaches -- the texture path's cost is per memory
  // INSTRUCTION, so instructions with most lanes idle waste it.  So the wave's 64 rows are flattened per field: every 16-byte
  // chunk of every row is one item, lane l takes items l, l + 64, ... (the row of an item: binary search over the rows'
  // first-item numbers in LDS), and every load / store instruction has all 64 lanes busy.  A field shorter than 16 bytes is
  // one item that travels in registers (8-byte load(s), 8 / 4 / 2 / 1-byte stores); a partial last chunk is served by the
  // overlapping 16 bytes that END at the field's end.  FQ_UNROLL items per lane are in flight.
  const uint32_t nrow = (uint32_t)((n - tile * ROWS_TILE) < ROWS_TILE ? (n - tile * ROWS_TILE) : ROWS_TILE);
  const uint32_t wrow0 = (uint32_t)w * WAVE;
  const uint32_t myrow = wrow0 + (uint32_t)lane;
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  auto chunks_of = [](uint32_t len) { return len >= 16 ? (len + 15u) >> 4 : (len ? 1u : 0u); };
  // items are numbered row by row and, inside a row, field by field: a batch of 64 x FQ_UNROLL items covers one contiguous
  // stretch of the text, so every line of it is fetched once (field-by-field passes over the wave's 64 rows re-read the
  // text after it had left the L2)
  uint32_t nch = 0;
  if (myrow < nrow) {
#pragma unroll
    for (int k = 0; k < 4; k++) if (c.val[k]) nch += chunks_of(s_len[k][myrow]);
  }
  uint32_t inc = nch;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) { const uint32_t o = __shfl_up(inc, d, WAVE); if (lane >= d) inc += o; }
  const uint32_t T = __builtin_amdgcn_readlane(inc, 63);
  s_cs[w][lane] = inc - nch;
  wave_sync();
  for (uint32_t i0 = 0; i0 < T; i0 += WAVE * FQ_UNROLL) {
    fq_u32x4 v[FQ_UNROLL];
    uint32_t it_len[FQ_UNROLL], it_cc[FQ_UNROLL];
    const uint8_t* it_src[FQ_UNROLL];
    uint8_t* it_dst[FQ_UNROLL];
#pragma unroll
    for (int q = 0; q < FQ_UNROLL; q++) {
      const uint32_t i = i0 + (uint32_t)q * WAVE + (uint32_t)lane;
      v[q] = fq_u32x4{0, 0, 0, 0};
      it_len[q] = 0; it_cc[q] = 0; it_src[q] = u; it_dst[q] = nullptr;
      if (i >= T) continue;
      uint32_t lo = 0, hi = WAVE - 1;   // the largest row whose first item number is <= i (empty rows share the next row's number)
#pragma unroll
      for (int step = 0; step < 6; step++) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (s_cs[w][mid] <= i) lo = mid; else hi = mid - 1;
      }
      const uint32_t r = wrow0 + lo;
      uint32_t j = i - s_cs[w][lo];
      int kf = -1;
      uint32_t len = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (!c.val[k] || kf >= 0) continue;
        const uint32_t lk = s_len[k][r], nk = chunks_of(lk);
        if (j < nk) { kf = k; len = lk; } else j -= nk;
      }
      if (kf < 0) continue;
      it_len[q] = len;
      it_src[q] = u + s_src[kf][r];
      it_dst[q] = c.val[kf] + s_dst[kf][r];
      if (len >= 16) { const uint32_t c0 = j * 16u; it_cc[q] = c0 + 16 <= len ? c0 : len - 16; v[q] = *(const fq_u32x4*)(it_src[q] + it_cc[q]); }
      else {
        const uint64_t t = ((const fq_u64*)it_src[q])->v;   // (the text is padded)
        v[q].x = (uint32_t)t; v[q].y = (uint32_t)(t >> 32);
        if (len > 8) { const uint64_t t2 = ((const fq_u64*)(it_src[q] + len - 8))->v; v[q].z = (uint32_t)t2; v[q].w = (uint32_t)(t2 >> 32); }
      }
    }
#pragma unroll
    for (int q = 0; q < FQ_UNROLL; q++) {
      const uint32_t len = it_len[q];
      if (!len) continue;
      uint8_t* dp = it_dst[q];
      if (len >= 16) { *(fq_u32x4*)(dp + it_cc[q]) = v[q]; continue; }
      uint64_t t = (uint64_t)v[q].x | ((uint64_t)v[q].y << 32);
      if (len >= 8) {
        ((fq_u64*)dp)->v = t;
        if (len > 8) ((fq_u64*)(dp + len - 8))->v = (uint64_t)v[q].z | ((uint64_t)v[q].w << 32);
      } else {
        uint32_t o = 0;
        if (len & 4) { ((fq_u32*)dp)->v = (uint32_t)t; t >>= 32; o = 4; }
        if (len & 2) { ((fq_u16*)(dp + o))->v = (uint16_t)t; t >>= 16; o += 2; }
        if (len & 1) dp[o] = (uint8_t)t;
      }
    }
  }
}
void launch_fastq_pass1(const uint8_t* u, uint64_t x0, uint64_t eof, const uint64_t* nl, uint64_t n_nl, uint64_t n_rec, FqCols c,
                        uint64_t* tile_sums, uint32_t* err, hipStream_t st) {
  if (!n_rec) return;
  const uint64_t n_tiles = (n_rec + ROWS_TILE - 1) / ROWS_TILE;
  hipLaunchKernelGGL(k_fastq_pass1, dim3((uint32_t)n_tiles), dim3(ROWS_TILE), 0, st, u, x0, eof, nl, n_nl, n_rec, c, n_tiles, tile_sums, err);
  if (c.want) launch_tile_scan(tile_sums, n_tiles, c.want, st);
}
void launch_fastq_pass2(const uint8_t* u, uint64_t x0, uint64_t eof, const uint64_t* nl, uint64_t n_nl, uint64_t n_rec, FqCols c,
                        uint32_t batch_size, uint32_t phase, const uint64_t* tile_sums, hipStream_t st) {
  if (!n_rec || !c.want) return;
  const uint64_t n_tiles = (n_rec + ROWS_TILE - 1) / ROWS_TILE;
  const uint64_t nb = (n_rec + phase + batch_size - 1) / batch_size;
  hipLaunchKernelGGL(k_fastq_batch_bases, dim3((uint32_t)nb), dim3(ROWS_TILE), 0, st, u, x0, eof, nl, n_nl, n_rec, c, batch_size, phase, n_tiles, tile_sums);
  hipLaunchKernelGGL(k_fastq_pass2, dim3((uint32_t)n_tiles), dim3(ROWS_TILE), 0, st, u, x0, eof, nl, n_nl, n_rec, c, batch_size, phase, n_tiles, tile_sums);
}

// number of records whose first byte lies before `limit_off` (ownership threshold): record r starts at
// x0 (r = 0) or nl[4r-1]+1.  Single thread binary search.
__global__ void k_fastq_count_owned(const uint64_t* __restrict__ nl, uint64_t n_nl, uint64_t x0, uint64_t eof, uint64_t limit_off,
                                    unsigned long long* result) {
  // candidate records: a record exists at index r if its start < eof
  uint64_t lo = 0, hi = n_nl / 4 + 2;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    uint64_t st;
    bool exists = true;
    if (mid == 0) st = x0;
    else if (4 * mid - 1 < n_nl) st = (nl[4 * mid - 1] & NL_POS) + 1;
    else { st = eof; exists = false; }
    if (exists && st < eof && st < limit_off) lo = mid + 1; else hi = mid;
  }
  result[0] = lo;
}
void launch_fastq_count_owned(const uint64_t* nl, uint64_t n_nl, uint64_t x0, uint64_t eof, uint64_t limit_off,
                              unsigned long long* result, hipStream_t st) {
  hipLaunchKernelGGL(k_fastq_count_owned, dim3(1), dim3(1), 0, st, nl, n_nl, x0, eof, limit_off, result);
}

// ---- generic range scatter: row i copies len_i bytes from u[src_i..] to dst[off64_i..] -----------------
// Row-centric variant: G lanes per row (G = 4 for short fields, 16 for reads), 64 / G rows in flight per
// wave; each lane moves 16-byte chunks, a partial last chunk is served by the overlapping 16 bytes that end at
// the row's end, rows shorter than 16 bytes are copied by their first lane.  No binary search, no LDS.
struct __attribute__((packed, aligned(1))) rs_u32x4 { uint32_t x, y, z, w; };
// RS_UNROLL rows per lane group are in flight at once (all first-chunk loads before the first store): one row at a time the
// kernel waited a full memory latency per row (the same finding as pass 2 of bam_rows.hip).
#ifndef RS_UNROLL
#define RS_UNROLL 4
#endif
template <int G>
__global__ __launch_bounds__(256) void k_scatter_ranges_rows(const uint8_t* __restrict__ u, const uint64_t* __restrict__ src,
                                                              uint64_t n, const uint64_t* __restrict__ off64,
                                                              uint8_t* __restrict__ dst) {
  constexpr uint32_t RPB = 256 / G;  // rows one pass of the workgroup covers
  const uint64_t r0 = (uint64_t)blockIdx.x * (RPB * RS_UNROLL) + threadIdx.x / G;
  const uint32_t sl = (uint32_t)(threadIdx.x % G);
  const uint8_t* s[RS_UNROLL];
  uint8_t* d[RS_UNROLL];
  uint32_t len[RS_UNROLL], cc[RS_UNROLL];
  rs_u32x4 v[RS_UNROLL];
#pragma unroll
  for (int q = 0; q < RS_UNROLL; q++) {
    const uint64_t r = r0 + (uint64_t)q * RPB;
    len[q] = 0; s[q] = u; d[q] = dst; cc[q] = 0;
    v[q] = rs_u32x4{0, 0, 0, 0};
    if (r >= n) continue;
    const uint64_t o = off64[r];
    len[q] = (uint32_t)(off64[r + 1] - o);
    s[q] = u + src[r];
    d[q] = dst + o;
    const uint32_t c = sl * 16;
    cc[q] = c + 16 <= len[q] ? c : len[q] - 16;   // (only used when len >= 16 and c < len)
    if (len[q] >= 16 && c < len[q]) v[q] = *(const rs_u32x4*)(s[q] + cc[q]);
  }
#pragma unroll
  for (int q = 0; q < RS_UNROLL; q++) {
    const uint32_t l = len[q];
    if (l < 16) {
      if (sl == 0) for (uint32_t k = 0; k < l; k++) d[q][k] = s[q][k];
      continue;
    }
    if (sl * 16 < l) *(rs_u32x4*)(d[q] + cc[q]) = v[q];
    for (uint32_t c = sl * 16 + G * 16; c < l; c += G * 16) {
      const uint32_t c2 = c + 16 <= l ? c : l - 16;
      *(rs_u32x4*)(d[q] + c2) = *(const rs_u32x4*)(s[q] + c2);
    }
  }
}
// Tiny rows (GT strings, CHROM, REF, ALT ...: a few bytes each): one lane per row, the row travels in a register -- one
// unaligned 8-byte load (the source buffers have slack), then exactly `len` bytes go out in 4 / 2 / 1-byte pieces, so
// neighbouring rows written by neighbouring lanes are never touched.  Longer rows loop over 8-byte chunks with an
// overlapping tail.  (The output-centric kernel spends a binary search per output byte on such rows.)
struct __attribute__((packed, aligned(1))) rs_u64 { uint64_t v; };
struct __attribute__((packed, aligned(1))) rs_u32 { uint32_t v; };
struct __attribute__((packed, aligned(1))) rs_u16 { uint16_t v; };
__global__ __launch_bounds__(256) void k_scatter_ranges_tiny(const uint8_t* __restrict__ u, const uint64_t* __restrict__ src, uint64_t n,
                                                              const uint64_t* __restrict__ off64, uint8_t* __restrict__ dst) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const uint64_t o = off64[r];
  const uint32_t len = (uint32_t)(off64[r + 1] - o);
  if (!len) return;
  const uint8_t* s = u + src[r];
  uint8_t* d = dst + o;
  if (len >= 8) {
    for (uint32_t c = 0; c < len; c += 8) {
      const uint32_t cc = c + 8 <= len ? c : len - 8;
      ((rs_u64*)(d + cc))->v = ((const rs_u64*)(s + cc))->v;
    }
    return;
  }
  uint64_t w = ((const rs_u64*)s)->v;
  uint32_t k = 0;
  if (len & 4) { ((rs_u32*)d)->v = (uint32_t)w; w >>= 32; k = 4; }
  if (len & 2) { ((rs_u16*)(d + k))->v = (uint16_t)w; w >>= 16; k += 2; }
  if (len & 1) d[k] = (uint8_t)w;
}
void launch_scatter_ranges(const uint8_t* u, const uint64_t* src, uint64_t n, const uint64_t* off64, uint8_t* dst, uint64_t total_bytes,
                           hipStream_t st) {
  if (!n) return;
  if (total_bytes < 16 * n) {
    hipLaunchKernelGGL(k_scatter_ranges_tiny, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, src, n, off64, dst);
    return;
  }
  // the average row length picks the shape: tiny fields (CHROM, REF, ALT ...) one row per lane (above), medium rows
  // 4 lanes each, reads 8 or 16 lanes each
  auto blocks = [&](uint64_t g) { const uint64_t rows_per_block = (256 / g) * RS_UNROLL; return dim3((uint32_t)((n + rows_per_block - 1) / rows_per_block)); };
  if (total_bytes < 48 * n)
    hipLaunchKernelGGL(k_scatter_ranges_rows<4>, blocks(4), dim3(256), 0, st, u, src, n, off64, dst);
  else if (total_bytes < 144 * n)   // e.g. 101-base reads: 7 of 8 lanes busy instead of 7 of 16
    hipLaunchKernelGGL(k_scatter_ranges_rows<8>, blocks(8), dim3(256), 0, st, u, src, n, off64, dst);
  else
    hipLaunchKernelGGL(k_scatter_ranges_rows<16>, blocks(16), dim3(256), 0, st, u, src, n, off64, dst);
}

}  // namespace bioscan
