// fastq_kernels.hip -- FASTQ record framing + field extract -> Arrow scatter (gfx950).
//
// Replaces the per-record loop of the reference's FastqExec
// (bio-format-fastq/src/physical_exec.rs:393-465 `batch_producer`, :184-248 resync) and
// noodles-fastq 0.23.0 `Reader::read_record` (un-vendored).  Byte work only.
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
__global__ __launch_bounds__(256) void k_nl_count(const uint8_t* __restrict__ u, uint64_t lo, uint64_t hi, uint32_t* __restrict__ cnt) {
  __shared__ uint32_t s_w[4];
  const uint64_t a = lo + (uint64_t)blockIdx.x * NL_CHUNK + (uint64_t)threadIdx.x * NL_PER_THREAD;
  uint64_t m[8];
  uint32_t n = a < hi ? nl_thread_masks(u, a, hi, m) : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) n += __shfl_down(n, d, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
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
void launch_nl_count(const uint8_t* u, uint64_t lo, uint64_t hi, uint32_t* cnt, hipStream_t st) {
  const uint64_t n = nl_chunks(lo, hi);
  if (!n) return;
  hipLaunchKernelGGL(k_nl_count, dim3((uint32_t)n), dim3(256), 0, st, u, lo, hi, cnt);
}
void launch_nl_write(const uint8_t* u, uint64_t lo, uint64_t hi, const uint64_t* base, uint64_t* nl, hipStream_t st) {
  const uint64_t n = nl_chunks(lo, hi);
  if (!n) return;
  hipLaunchKernelGGL(k_nl_write, dim3((uint32_t)n), dim3(256), 0, st, u, lo, hi, base, nl);
}

// ---- record fields ------------------------------------------------------------------------------------
// Record r spans lines 4r..4r+3 counted from x0 (the first record start).  nl[] holds the newline
// positions >= x0 in order; a last line without '\n' ends at `eof` (only legal at the end of the data).
// Outputs per record: source offset + length of name, description, sequence, quality; description
// validity (NULL when empty, physical_exec.rs:430-434); err: 1 = missing '@', 2 = missing '+'.
// bit 8k+7 set iff byte k of w equals c (exact SWAR zero-byte test)
__device__ __forceinline__ uint64_t eq_mask8(uint64_t w, uint64_t c8) {
  const uint64_t x = w ^ c8;
  const uint64_t t = (x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full;
  return ~(t | x | 0x7F7F7F7F7F7F7F7Full);
}
__global__ __launch_bounds__(256) void k_fastq_fields(const uint8_t* __restrict__ u, uint64_t x0, uint64_t eof,
                                                       const uint64_t* __restrict__ nl, uint64_t n_nl, uint64_t n_rec,
                                                       FastqCols c, uint32_t* err) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = r < n_rec;
  bool dvalid = false;
  if (act) {
    // the five index entries around the record: nl[4r-1] .. nl[4r+3]
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
    if (first[0] != '@') atomicExch(err, 1u);
    if (s[2] < eof && first[2] != '+') atomicExch(err, 2u);
    // name = header up to the first space or tab, found eight bytes at a time
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
    const uint64_t name_len = sp - d0;
    const uint64_t desc_off = sp < e[0] ? sp + 1 : e[0];
    const uint64_t desc_len = e[0] - desc_off;
    dvalid = desc_len != 0;
    if (c.src_name) { c.src_name[r] = d0; c.len_name[r] = (uint32_t)name_len; }
    if (c.src_desc) { c.src_desc[r] = desc_off; c.len_desc[r] = (uint32_t)desc_len; }
    if (c.src_seq) { c.src_seq[r] = s[1]; c.len_seq[r] = (uint32_t)(e[1] - s[1]); }
    if (c.src_qual) { c.src_qual[r] = s[3]; c.len_qual[r] = (uint32_t)(e[3] - s[3]); }
  }
  if (c.v_desc) {
    const unsigned long long m = __ballot(dvalid);
    if ((threadIdx.x & 63) == 0 && act) c.v_desc[r >> 6] = m;
  }
}
void launch_fastq_fields(const uint8_t* u, uint64_t x0, uint64_t eof, const uint64_t* nl, uint64_t n_nl, uint64_t n_rec,
                         FastqCols c, uint32_t* err, hipStream_t st) {
  if (!n_rec) return;
  hipLaunchKernelGGL(k_fastq_fields, dim3((uint32_t)((n_rec + 255) / 256)), dim3(256), 0, st, u, x0, eof, nl, n_nl, n_rec, c, err);
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
template <int G>
__global__ __launch_bounds__(256) void k_scatter_ranges_rows(const uint8_t* __restrict__ u, const uint64_t* __restrict__ src,
                                                              uint64_t n, const uint64_t* __restrict__ off64,
                                                              uint8_t* __restrict__ dst) {
  const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t r = t / G;
  const uint32_t sl = (uint32_t)(t % G);
  if (r >= n) return;
  const uint64_t o = off64[r];
  const uint32_t len = (uint32_t)(off64[r + 1] - o);
  const uint8_t* s = u + src[r];
  uint8_t* d = dst + o;
  if (len < 16) {
    if (sl == 0) for (uint32_t k = 0; k < len; k++) d[k] = s[k];
    return;
  }
  for (uint32_t c = sl * 16; c < len; c += G * 16) {
    const uint32_t cc = c + 16 <= len ? c : len - 16;
    *(rs_u32x4*)(d + cc) = *(const rs_u32x4*)(s + cc);
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
  if (total_bytes < 48 * n)
    hipLaunchKernelGGL(k_scatter_ranges_rows<4>, dim3((uint32_t)((n * 4 + 255) / 256)), dim3(256), 0, st, u, src, n, off64, dst);
  else if (total_bytes < 144 * n)   // e.g. 101-base reads: 7 of 8 lanes busy instead of 7 of 16
    hipLaunchKernelGGL(k_scatter_ranges_rows<8>, dim3((uint32_t)((n * 8 + 255) / 256)), dim3(256), 0, st, u, src, n, off64, dst);
  else
    hipLaunchKernelGGL(k_scatter_ranges_rows<16>, dim3((uint32_t)((n * 16 + 255) / 256)), dim3(256), 0, st, u, src, n, off64, dst);
}

}  // namespace bioscan
