// vcf_kernels.hip -- gfx950 kernels of the VCF text path (K9 in DESIGN.md): delimiter index, line keys,
// row selection, core / INFO / FORMAT field extraction into Arrow buffers, list UDFs.
//
// Replaces, per record, noodles-vcf's lazy `Record` accessors driven by
// bio-format-vcf/src/physical_exec.rs:991-1116 (sequential) / :2858-2999 (indexed), load_infos_single_pass
// :544-640, MultiSampleFormatBuilder::append_record :1634-1828, load_formats_single_pass :2265-2443 and the
// list UDFs udfs.rs:67-110, :606-650.
//
// Text layout: `u` holds the decoded bytes of the scanned range; every offset below is relative to u.
// The delimiter index is two sorted position arrays ('\n' and '\t') plus, for every newline, the number
// of tabs before it, so field f of line i is found with two loads instead of a byte scan.
#include "vcf_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>

namespace bioscan {

constexpr int WAVE = 64;
constexpr int DL_CHUNK = 16384;
constexpr int DL_PER_THREAD = 64;
struct __attribute__((packed, aligned(1))) dl_u64 { uint64_t v; };
struct __attribute__((packed, aligned(1))) dl_u64x2 { uint64_t a, b; };
struct __attribute__((packed, aligned(1))) dl_u32 { uint32_t v; };

// bit 8k+7 of the result is set iff byte k of w equals c (c replicated in `pat`)
__device__ __forceinline__ uint64_t eq_mask8(uint64_t w, uint64_t pat) {
  const uint64_t x = w ^ pat;
  const uint64_t t = (x & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full;
  return ~(t | x | 0x7F7F7F7F7F7F7F7Full);
}
__device__ __forceinline__ void dl_thread_masks(const uint8_t* __restrict__ u, uint64_t a, uint64_t hi, uint64_t mn[8], uint64_t mt[8],
                                                uint32_t* n_nl, uint32_t* n_tab) {
  uint32_t cn = 0, ct = 0;
  uint64_t wv[8];
  if (a + 64 <= hi) {  // whole 64 bytes inside the range: four 16-byte loads
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const dl_u64x2 v = *(const dl_u64x2*)(u + a + 16 * k);
      wv[2 * k] = v.a; wv[2 * k + 1] = v.b;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint64_t p = a + 8 * k;
      uint64_t w = 0;
      if (p + 8 <= hi) w = ((const dl_u64*)(u + p))->v;
      else if (p < hi) { for (uint64_t q = p; q < hi; q++) w |= (uint64_t)u[q] << (8 * (q - p)); }
      wv[k] = w;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint64_t p = a + 8 * k;
    const uint64_t w = wv[k];
    uint64_t a_n = 0, a_t = 0;
    if (p < hi) {
      a_n = eq_mask8(w, 0x0A0A0A0A0A0A0A0Aull);
      a_t = eq_mask8(w, 0x0909090909090909ull);
      if (p + 8 > hi) {  // bytes past hi read as 0: never '\n' or '\t'
      }
    }
    mn[k] = a_n;
    mt[k] = a_t;
    cn += (uint32_t)__popcll(a_n);
    ct += (uint32_t)__popcll(a_t);
  }
  *n_nl = cn;
  *n_tab = ct;
}

__global__ __launch_bounds__(256) void k_delim_count(const uint8_t* __restrict__ u, uint64_t lo, uint64_t hi,
                                                      uint32_t* __restrict__ cnt_nl, uint32_t* __restrict__ cnt_tab) {
  __shared__ uint32_t s_n[4], s_t[4];
  const uint64_t a = lo + (uint64_t)blockIdx.x * DL_CHUNK + (uint64_t)threadIdx.x * DL_PER_THREAD;
  uint64_t mn[8], mt[8];
  uint32_t n = 0, t = 0;
  if (a < hi) dl_thread_masks(u, a, hi, mn, mt, &n, &t);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    n += __shfl_down(n, d, 64);
    t += __shfl_down(t, d, 64);
  }
  if ((threadIdx.x & 63) == 0) { s_n[threadIdx.x >> 6] = n; s_t[threadIdx.x >> 6] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    cnt_nl[blockIdx.x] = s_n[0] + s_n[1] + s_n[2] + s_n[3];
    cnt_tab[blockIdx.x] = s_t[0] + s_t[1] + s_t[2] + s_t[3];
  }
}

__global__ __launch_bounds__(256) void k_delim_write(const uint8_t* __restrict__ u, uint64_t lo, uint64_t hi,
                                                      const uint64_t* __restrict__ base_nl, const uint64_t* __restrict__ base_tab,
                                                      uint64_t* __restrict__ nl, uint64_t* __restrict__ nl_tabs,
                                                      uint64_t* __restrict__ tab) {
  __shared__ uint32_t s_n[4], s_t[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t a = lo + (uint64_t)blockIdx.x * DL_CHUNK + (uint64_t)threadIdx.x * DL_PER_THREAD;
  uint64_t mn[8], mt[8];
  uint32_t n = 0, t = 0;
  if (a < hi) dl_thread_masks(u, a, hi, mn, mt, &n, &t);
  uint32_t in_n = n, in_t = t;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t on = __shfl_up(in_n, d, 64), ot = __shfl_up(in_t, d, 64);
    if (lane >= d) { in_n += on; in_t += ot; }
  }
  if (lane == 63) { s_n[wv] = in_n; s_t[wv] = in_t; }
  __syncthreads();
  uint32_t wb_n = 0, wb_t = 0;
  for (int i = 0; i < wv; i++) { wb_n += s_n[i]; wb_t += s_t[i]; }
  uint64_t on = base_nl[blockIdx.x] + wb_n + (in_n - n);
  uint64_t ot = base_tab[blockIdx.x] + wb_t + (in_t - t);
  if (n | t) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      uint64_t mk = mn[k];
      while (mk) {
        const int bit = __builtin_ctzll(mk);
        nl[on] = a + 8 * k + (uint64_t)(bit >> 3);
        nl_tabs[on] = ot + (uint64_t)__popcll(mt[k] & ((1ull << bit) - 1ull));
        on++;
        mk &= mk - 1;
      }
      uint64_t tk = mt[k];
      while (tk) {
        const int bit = __builtin_ctzll(tk);
        tab[ot++] = a + 8 * k + (uint64_t)(bit >> 3);
        tk &= tk - 1;
      }
    }
  }
}

uint64_t vcf_delim_chunks(uint64_t lo, uint64_t hi) { return hi > lo ? (hi - lo + DL_CHUNK - 1) / DL_CHUNK : 0; }
void launch_vcf_delim_count(const uint8_t* u, uint64_t lo, uint64_t hi, uint32_t* cnt_nl, uint32_t* cnt_tab, hipStream_t st) {
  const uint64_t n = vcf_delim_chunks(lo, hi);
  if (!n) return;
  hipLaunchKernelGGL(k_delim_count, dim3((uint32_t)n), dim3(256), 0, st, u, lo, hi, cnt_nl, cnt_tab);
}
void launch_vcf_delim_write(const uint8_t* u, uint64_t lo, uint64_t hi, const uint64_t* base_nl, const uint64_t* base_tab,
                            uint64_t* nl, uint64_t* nl_tabs, uint64_t* tab, hipStream_t st) {
  const uint64_t n = vcf_delim_chunks(lo, hi);
  if (!n) return;
  hipLaunchKernelGGL(k_delim_write, dim3((uint32_t)n), dim3(256), 0, st, u, lo, hi, base_nl, base_tab, nl, nl_tabs, tab);
}

// ---- line table -----------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t line_start(const VcfLines& L, uint64_t i) { return i == 0 ? L.x0 : L.nl[i - 1] + 1; }
__device__ __forceinline__ uint64_t line_end_raw(const VcfLines& L, uint64_t i) { return i < L.n_nl ? L.nl[i] : L.hi; }
__device__ __forceinline__ uint64_t line_tb(const VcfLines& L, uint64_t i) { return i == 0 ? 0 : L.nl_tabs[i - 1]; }
__device__ __forceinline__ uint64_t line_te(const VcfLines& L, uint64_t i) { return i < L.n_nl ? L.nl_tabs[i] : L.n_tab; }
// span of field f of line i; false when the line has fewer fields
__device__ __forceinline__ bool field_span(const VcfLines& L, const uint8_t* u, uint64_t i, uint32_t f, uint64_t* a, uint64_t* b) {
  const uint64_t tb = line_tb(L, i), te = line_te(L, i);
  const uint64_t nt = te - tb;
  if (f > nt) return false;
  *a = f == 0 ? line_start(L, i) : L.tab[tb + f - 1] + 1;
  if (f < nt) *b = L.tab[tb + f];
  else {
    uint64_t e = line_end_raw(L, i);
    if (e > *a && u[e - 1] == '\r') e--;
    *b = e;
  }
  return true;
}

// ---- number parsing ---------------------------------------------------------------------------------------
// i32 from [+-]?[0-9]+ ; 0 ok, 1 malformed / out of range
__device__ int parse_i32_text(const uint8_t* p, uint32_t len, int32_t* out) {
  uint32_t i = 0;
  bool neg = false;
  if (i < len && (p[i] == '+' || p[i] == '-')) { neg = p[i] == '-'; i++; }
  if (i >= len) return 1;
  int64_t v = 0;
  for (; i < len; i++) {
    const uint32_t d = (uint32_t)p[i] - '0';
    if (d > 9) return 1;
    v = v * 10 + d;
    if (v > 2147483648ll) return 1;
  }
  if (neg) v = -v;
  if (v > 2147483647ll) return 1;
  *out = (int32_t)v;
  return 0;
}
// POS: noodles parses a usize (an error beyond 2^64 - 1) and the reference casts it (`get() as u32`,
// bio-format-vcf/src/physical_exec.rs:762, 2875): a position that does not fit wraps.  *wide: the value needed more than 32 bits.
__device__ int parse_u32_pos(const uint8_t* p, uint32_t len, uint32_t* out, bool* wide, bool* zero) {
  uint32_t i = (len && p[0] == '+') ? 1u : 0u;   // (usize::from_str takes a leading '+')
  if (i >= len) return 1;
  uint64_t v = 0;
  for (; i < len; i++) {
    const uint32_t d = (uint32_t)p[i] - '0';
    if (d > 9) return 1;
    if (v > (0xFFFFFFFFFFFFFFFFull - d) / 10ull) return 1;
    v = v * 10 + d;
  }
  *out = (uint32_t)v;
  *wide = v > 0xFFFFFFFFull;
  *zero = v == 0;
  return 0;
}

__device__ const double P10D[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11,
                                    1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
__device__ const uint64_t P10U[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull,
                                      100000000ull, 1000000000ull, 10000000000ull, 100000000000ull, 1000000000000ull,
                                      10000000000000ull, 100000000000000ull, 1000000000000000ull, 10000000000000000ull,
                                      100000000000000000ull, 1000000000000000000ull, 10000000000000000000ull};
__device__ __forceinline__ bool ci_eq(const uint8_t* p, uint32_t len, const char* s) {
  uint32_t i = 0;
  for (; s[i]; i++) {
    if (i >= len) return false;
    uint8_t c = p[i];
    if (c >= 'A' && c <= 'Z') c += 32;
    if (c != (uint8_t)s[i]) return false;
  }
  return i == len;
}
// f64 -> f32, round to nearest even, on the bit patterns: results below 2^-126 come out as the exact subnormal whatever
// the denormal mode of the float pipeline is (INFO fields carry p-values like 3.1e-42).
__device__ __forceinline__ float f64_to_f32_rne(double d) {
  const uint64_t b = (uint64_t)__double_as_longlong(d);
  const uint32_t sign = (uint32_t)(b >> 63) << 31;
  const int e = (int)((b >> 52) & 0x7FF) - 1023;
  if (e >= -126) return (float)d;                       // normal range (or inf / nan / zero handled by the conversion)
  if (((b >> 52) & 0x7FF) == 0) return __uint_as_float(sign);  // zero / f64 subnormal: far below 2^-149
  const uint64_t mant = (b & ((1ull << 52) - 1)) | (1ull << 52);
  const int s = -(e + 97);                              // value = mant * 2^(e - 52) = q * 2^-149 with q = mant >> s
  if (s > 54) return __uint_as_float(sign);             // below half the smallest subnormal
  const uint64_t q = mant >> s, rest = mant & ((1ull << s) - 1), half = 1ull << (s - 1);
  const uint64_t r = q + ((rest > half || (rest == half && (q & 1))) ? 1 : 0);
  return __uint_as_float(sign | (uint32_t)r);           // r == 2^23 is the smallest normal: the same encoding
}
// ---- exact slow path of the float parser: big integers ------------------------------------------------------------
// Reached only when the fast paths below cannot prove the rounding (a literal within one part in 2^52 of the midpoint of two
// neighbouring f32 values -- the midpoints themselves have up to ~150 significant digits).  The literal's digits (the first
// BIG_DIGITS significant ones exactly, the rest as a sticky bit) and the midpoint M x 2^q are compared as integers:
// D x 10^k <=> M x 2^q after the powers of five and two have been moved to the side where they are non-negative.
constexpr int BIG_N = 48;        // 1536 bits: 300 digits (997 bits) x 2^(<= 350) or 2^53 x 5^(<= 346) x 2^(<= 350)
constexpr int BIG_DIGITS = 300;
struct Big { uint32_t w[BIG_N]; };
__device__ __noinline__ void big_mul_add(Big* a, uint32_t mul, uint32_t add) {
  uint64_t carry = add;
  for (int i = 0; i < BIG_N; i++) { const uint64_t t = (uint64_t)a->w[i] * mul + carry; a->w[i] = (uint32_t)t; carry = t >> 32; }
}
__device__ __noinline__ void big_shl(Big* a, uint32_t bits) {
  const int ws = (int)(bits >> 5), bs = (int)(bits & 31);
  for (int i = BIG_N - 1; i >= 0; i--) {
    uint32_t v = 0;
    if (i - ws >= 0) v = a->w[i - ws] << bs;
    if (bs && i - ws - 1 >= 0) v |= a->w[i - ws - 1] >> (32 - bs);
    a->w[i] = v;
  }
}
__device__ __noinline__ int big_cmp(const Big* a, const Big* b) {
  for (int i = BIG_N - 1; i >= 0; i--) if (a->w[i] != b->w[i]) return a->w[i] < b->w[i] ? -1 : 1;
  return 0;
}
__device__ void big_pow5(Big* a, int64_t k) {
  while (k >= 13) { big_mul_add(a, 1220703125u, 0); k -= 13; }   // 5^13
  uint32_t r = 1;
  for (; k > 0; k--) r *= 5u;
  if (r != 1) big_mul_add(a, r, 0);
}
// sign of (D x 10^k [+ a little when sticky]) - mid, mid > 0 finite
__device__ __noinline__ int big_cmp_mid(const Big* D, int64_t k, bool sticky, double mid) {
  int e;
  const double fr = frexp(mid, &e);
  const uint64_t M = (uint64_t)ldexp(fr, 53);
  const int64_t q = (int64_t)e - 53;
  Big A = *D, B;
  for (int i = 0; i < BIG_N; i++) B.w[i] = 0;
  B.w[0] = (uint32_t)M; B.w[1] = (uint32_t)(M >> 32);
  int64_t sh;  // A x 2^sh <=> B
  if (k >= 0) { big_pow5(&A, k); sh = k - q; } else { big_pow5(&B, -k); sh = k - q; }
  if (sh >= 0) big_shl(&A, (uint32_t)sh); else big_shl(&B, (uint32_t)(-sh));
  const int c = big_cmp(&A, &B);
  return c != 0 ? c : (sticky ? 1 : 0);
}
// |value| of the literal at p (already known to be well formed, finite, non-zero), correctly rounded; `approx` is any f32
// within a few ulps of it
__device__ __noinline__ float parse_f32_exact(const uint8_t* p, uint32_t len, float approx) {
  Big D;
  for (int i = 0; i < BIG_N; i++) D.w[i] = 0;
  uint32_t i = 0;
  if (i < len && (p[i] == '+' || p[i] == '-')) i++;
  int nd = 0;
  int64_t k = 0;
  bool dot = false, sticky = false, lead = true;
  for (; i < len; i++) {
    const uint8_t c = p[i];
    const uint32_t d = (uint32_t)c - '0';
    if (d <= 9) {
      if (lead && d == 0) { if (dot) k--; continue; }
      lead = false;
      if (nd < BIG_DIGITS) { big_mul_add(&D, 10u, d); nd++; if (dot) k--; }
      else { if (d) sticky = true; if (!dot) k++; }
    } else if (c == '.') dot = true;
    else break;
  }
  if (i < len && (p[i] == 'e' || p[i] == 'E')) {
    i++;
    bool eneg = false;
    if (i < len && (p[i] == '+' || p[i] == '-')) { eneg = p[i] == '-'; i++; }
    int64_t ex = 0;
    for (; i < len; i++) { const uint32_t d = (uint32_t)p[i] - '0'; if (d > 9) break; if (ex < 100000) ex = ex * 10 + d; }
    k += eneg ? -ex : ex;
  }
  // candidates: approx and its two neighbours on the f32 grid (2^128 stands in for infinity)
  const double TOP = 3.402823669209384634633746074317682e38;  // 2^128
  auto as_d = [&](float f) { return isinf(f) ? TOP : (double)f; };
  const uint32_t ab = __float_as_uint(approx);
  const float c0 = approx;
  const float lo = ab ? __uint_as_float(ab - 1) : 0.0f;                  // previous float (approx >= 0)
  const float hi = isinf(approx) ? approx : __uint_as_float(ab + 1);      // next float (0x7F7FFFFF + 1 = +inf)
  auto even = [](float f) { return (__float_as_uint(f) & 1u) == 0u; };
  if (ab) {
    const double mid1 = 0.5 * (as_d(lo) + as_d(c0));
    const int c1 = big_cmp_mid(&D, k, sticky, mid1);
    if (c1 < 0) return lo;
    if (c1 == 0) return even(lo) ? lo : c0;
  }
  if (isinf(c0)) return c0;
  const double mid2 = 0.5 * (as_d(c0) + as_d(hi));
  const int c2 = big_cmp_mid(&D, k, sticky, mid2);
  if (c2 < 0) return c0;
  if (c2 == 0) return even(c0) ? c0 : hi;
  return hi;
}

// Correctly rounded decimal -> f32 (what Rust's `str::parse::<f32>` returns).  0 ok; 1 malformed.  Short literals are rounded
// once from exact integer arithmetic, the others from an f64 estimate with an error bound; when the bound straddles the midpoint
// of two f32 values the digits are compared with that midpoint as big integers (parse_f32_exact), so the result is never a guess.
__device__ int parse_f32_text(const uint8_t* p, uint32_t len, float* out) {
  uint32_t i = 0;
  bool neg = false;
  if (i < len && (p[i] == '+' || p[i] == '-')) { neg = p[i] == '-'; i++; }
  if (ci_eq(p + i, len - i, "inf") || ci_eq(p + i, len - i, "infinity")) { *out = neg ? -__builtin_huge_valf() : __builtin_huge_valf(); return 0; }
  if (ci_eq(p + i, len - i, "nan")) { *out = __builtin_nanf(""); return 0; }
  uint64_t m = 0;
  int nd = 0;
  int64_t e10 = 0;
  bool any = false, dot = false, sticky = false;
  for (; i < len; i++) {
    const uint8_t c = p[i];
    const uint32_t d = (uint32_t)c - '0';
    if (d <= 9) {
      any = true;
      if (nd < 19) {
        m = m * 10 + d;
        if (m) nd++;
        if (dot) e10--;
      } else {
        if (d) sticky = true;
        if (!dot) e10++;
      }
    } else if (c == '.') {
      if (dot) return 1;
      dot = true;
    } else break;
  }
  if (!any) return 1;
  if (i < len && (p[i] == 'e' || p[i] == 'E')) {
    i++;
    bool eneg = false;
    if (i < len && (p[i] == '+' || p[i] == '-')) { eneg = p[i] == '-'; i++; }
    if (i >= len) return 1;
    int64_t ex = 0;
    for (; i < len; i++) {
      const uint32_t d = (uint32_t)p[i] - '0';
      if (d > 9) return 1;
      if (ex < 100000) ex = ex * 10 + d;
    }
    e10 += eneg ? -ex : ex;
  }
  if (i != len) return 1;
  float r;
  bool unsure = false;
  if (m == 0) r = 0.0f;
  else if (e10 + nd > 40) r = __builtin_huge_valf();
  else if (e10 + nd < -50) r = 0.0f;
  else if (!sticky && e10 >= 0 && e10 <= 19 && m <= 0xFFFFFFFFFFFFFFFFull / P10U[e10]) {
    // exact integer: one rounding, by the u64 -> f32 conversion
    r = (float)(m * P10U[e10]);
  } else if (e10 < 0 && e10 >= -19) {
    // exact quotient m / 10^k: 33 quotient bits by restoring division on normalised operands, sticky remainder
    const int lz = __builtin_clzll(m);
    const uint64_t mn = m << lz;
    const uint64_t d = P10U[-e10];
    const int dz = __builtin_clzll(d);
    const uint64_t dn = d << dz;
    uint64_t q = mn >= dn ? 1 : 0;
    uint64_t rem = mn >= dn ? mn - dn : mn;
    for (int k = 0; k < 32; k++) {
      const bool top = rem >> 63;
      rem <<= 1;
      q <<= 1;
      if (top || rem >= dn) { rem -= dn; q |= 1; }
    }
    unsure = sticky && (q & 0x7F) == 0x7F;  // dropped digits could carry into the rounding position
    if (rem || sticky) q |= 1ull;
    r = ldexpf((float)q, dz - lz - 32);  // m >= 1 and 10^k <= 10^19: never below 1e-19, far from the subnormal range
  } else {
    double d = (double)m;
    int64_t e = e10;
    double kerr = sticky || m >= (1ull << 53) ? 2.0 : 0.5;
    while (e > 22) { d *= 1e22; e -= 22; kerr += 1.0; }
    while (e < -22) { d /= 1e22; e += 22; kerr += 1.0; }
    if (e >= 0) d *= P10D[e]; else d /= P10D[-e];
    kerr += 0.5;
    const double eps = (kerr + 0.5) * 2.220446049250313e-16;
    const float flo = f64_to_f32_rne(d * (1.0 - eps)), fhi = f64_to_f32_rne(d * (1.0 + eps));
    r = flo;
    unsure = __float_as_uint(flo) != __float_as_uint(fhi);
  }
  *out = neg ? -r : r;
  return unsure ? 2 : 0;
}

__device__ __forceinline__ void set_err(uint32_t* err, uint32_t code) { atomicCAS(err, 0u, code); }
// A literal the fast paths could not round with certainty (parse_f32_text returned 2 and a provisional value): the cell is
// queued behind the error word and k_f32_fix rewrites it from the exact comparison after the kernel -- the big-integer code
// stays out of the hot kernels (inlined or called, it cost them 80 .. 250 registers).
__device__ void f32_defer(uint32_t* err, const uint8_t* p, uint32_t len, void* dst, uint32_t as_f64, float approx) {
  const uint32_t idx = atomicAdd(&err[1], 1u);
  if (idx >= F32_FIX_CAP) { set_err(err, VERR_FLOAT_PRECISION); return; }
  F32Fix* e = (F32Fix*)(err + 4) + idx;
  e->p = p; e->dst = dst; e->len = len; e->as_f64 = as_f64; e->approx = __float_as_uint(approx); e->pad = 0;
}
__global__ __launch_bounds__(64) void k_f32_fix(uint32_t* err) {
  const uint32_t n = err[1] < F32_FIX_CAP ? err[1] : F32_FIX_CAP;
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  if (t >= n) return;
  const F32Fix e = ((const F32Fix*)(err + 4))[t];
  const float a = fabsf(__uint_as_float(e.approx));
  float f = parse_f32_exact(e.p, e.len, a);
  if (e.len && e.p[0] == '-') f = -f;
  if (e.as_f64) *(double*)e.dst = (double)f; else *(float*)e.dst = f;
}
void launch_f32_fix(uint32_t* err, uint32_t n_queued, hipStream_t st) {
  const uint32_t n = n_queued < F32_FIX_CAP ? n_queued : F32_FIX_CAP;
  if (n) hipLaunchKernelGGL(k_f32_fix, dim3((n + 63) / 64), dim3(64), 0, st, err);
}
__device__ __forceinline__ int hexval(uint32_t c) {
  if (c - '0' <= 9u) return (int)(c - '0');
  c |= 0x20u;
  if (c - 'a' <= 5u) return (int)(c - 'a') + 10;
  return -1;
}
// length of [p, p+l) after percent-decoding (noodles decodes INFO / FORMAT strings; malformed escapes stay as they are).
// Sets *pct when an escape was decoded.  An escape may produce a byte >= 0x80: the decoded value then has to be valid
// UTF-8 as a whole (percent_decode(..).decode_utf8() in noodles), otherwise the record is an error.
__device__ uint32_t pct_decoded_len(const uint8_t* p, uint32_t l, bool* pct, uint32_t* err) {
  uint32_t out = 0;
  bool high = false;
  for (uint32_t k = 0; k < l; k++, out++) {
    if (p[k] == '%' && k + 2 < l) {
      const int h = hexval(p[k + 1]), lo = hexval(p[k + 2]);
      if (h >= 0 && lo >= 0) {
        if (h >= 8) high = true;
        *pct = true;
        k += 2;
      }
    }
  }
  if (high) {
    // second walk: UTF-8 validation of the decoded bytes (Unicode 15 table 3-7: no overlongs, no surrogates, <= U+10FFFF)
    uint32_t need = 0, lo_b = 0x80, hi_b = 0xBF;
    bool ok = true;
    for (uint32_t k = 0; k < l && ok; k++) {
      uint32_t c = p[k];
      if (c == '%' && k + 2 < l) {
        const int h = hexval(p[k + 1]), lo = hexval(p[k + 2]);
        if (h >= 0 && lo >= 0) { c = (uint32_t)(h * 16 + lo); k += 2; }
      }
      if (need) {
        ok = c >= lo_b && c <= hi_b;
        need--; lo_b = 0x80; hi_b = 0xBF;
      } else if (c < 0x80) {
      } else if (c >= 0xC2 && c <= 0xDF) { need = 1; }
      else if (c == 0xE0) { need = 2; lo_b = 0xA0; }
      else if ((c >= 0xE1 && c <= 0xEC) || c == 0xEE || c == 0xEF) { need = 2; }
      else if (c == 0xED) { need = 2; hi_b = 0x9F; }
      else if (c == 0xF0) { need = 3; lo_b = 0x90; }
      else if (c >= 0xF1 && c <= 0xF3) { need = 3; }
      else if (c == 0xF4) { need = 3; hi_b = 0x8F; }
      else ok = false;
    }
    if (!ok || need) set_err(err, VERR_PERCENT);
  }
  return out;
}

// ---- values of keys the scan has no column for ---------------------------------------------------------------------
// A kernel that looks many keys up copies the table to LDS first (a header declares a few dozen keys: <= 128 slots of 16 bytes);
// every lookup is then an LDS read instead of a dependent L2 round trip per INFO entry.  Called by all threads of the block.
constexpr uint32_t VCF_TT_LDS_SLOTS = 128;
__device__ __forceinline__ VcfTypeTable stage_type_table(VcfTypeTable T, VcfTypeSlot* s_slots) {
  if (T.slots != nullptr && T.mask < VCF_TT_LDS_SLOTS) {
    for (uint32_t i = threadIdx.x; i <= T.mask; i += blockDim.x) s_slots[i] = T.slots[i];
    __syncthreads();
    T.slots = s_slots;
  }
  return T;
}
__device__ uint32_t type_lookup(const VcfTypeTable& T, const uint8_t* k, uint32_t kl, uint32_t* sel = nullptr) {
  if (sel) *sel = 0;
  if (T.slots == nullptr) return T.miss_kind;
  uint64_t k8 = ((const dl_u64*)k)->v;   // (reads past the key stay inside the text buffer's slack)
  if (kl < 8) k8 &= (1ull << (8 * kl)) - 1ull;
  for (uint32_t h = vcf_key_hash(k8, kl) & T.mask;; h = (h + 1) & T.mask) {
    const VcfTypeSlot sl = T.slots[h];
    if (sl.len_kind == 0) return T.miss_kind;
    if (sl.key8 == k8 && (sl.len_kind & 0xFFFFFFu) == kl) {
      bool eq = true;
      for (uint32_t i = 8; i < kl && eq; i++) eq = T.keys[(sl.off & 0xFFFFu) + i] == k[i];
      if (eq) { if (sel) *sel = sl.off >> 16; return (sl.len_kind >> 24) - 1u; }
    }
  }
}
// a float literal Rust's `str::parse::<f32>` takes (the grammar of parse_f32_text, without the value)
__device__ bool f32_text_ok(const uint8_t* p, uint32_t len) {
  uint32_t i = (len && (p[0] == '+' || p[0] == '-')) ? 1u : 0u;
  if (ci_eq(p + i, len - i, "inf") || ci_eq(p + i, len - i, "infinity") || ci_eq(p + i, len - i, "nan")) return true;
  bool any = false, dot = false;
  for (; i < len; i++) {
    const uint32_t c = p[i];
    if (c - '0' <= 9u) any = true;
    else if (c == '.') { if (dot) return false; dot = true; }
    else break;
  }
  if (!any) return false;
  if (i < len && (p[i] == 'e' || p[i] == 'E')) {
    i++;
    if (i < len && (p[i] == '+' || p[i] == '-')) i++;
    if (i >= len) return false;
    for (; i < len; i++) if ((uint32_t)p[i] - '0' > 9u) return false;
  }
  return i == len;
}
// genotype text as noodles walks it: an optional leading phasing mark, then alleles ('.' or digits) joined by '/' or '|'
__device__ bool gt_text_ok(const uint8_t* p, uint32_t l) {
  uint32_t k = (l && (p[0] == '/' || p[0] == '|')) ? 1u : 0u;
  if (k >= l) return false;
  uint32_t tl = 0;
  bool isdot = false;
  for (; k < l; k++) {
    const uint32_t ch = p[k];
    if (ch == '/' || ch == '|') { if (tl == 0) return false; tl = 0; isdot = false; }
    else if (ch == '.') { if (tl) return false; isdot = true; tl = 1; }
    else if (ch - '0' <= 9u) { if (isdot) return false; tl++; }
    else return false;
  }
  return tl != 0;
}
// The common well-formed values in one 8-byte word: 1 .. 8 decimal digits (always an i32), or 1 .. 8 characters of digits with at
// most one '.' and at least one digit (a float literal without sign and exponent).  Anything else takes the parsers' road.
__device__ __forceinline__ bool swar_plain_number(const uint8_t* p, uint32_t l, bool allow_dot) {
  if (l == 0 || l > 8) return false;
  uint64_t w = ((const dl_u64*)p)->v;   // (reads past the value stay inside the text buffer's slack)
  if (l < 8) { const uint64_t keep = (1ull << (8 * l)) - 1ull; w = (w & keep) | (0x3030303030303030ull & ~keep); }
  uint32_t ndot = 0;
  if (allow_dot) {
    const uint64_t dm = eq_mask8(w, 0x2E2E2E2E2E2E2E2Eull);
    ndot = (uint32_t)__popcll(dm);
    w ^= (dm >> 7) * 0x1Eull;           // '.' -> '0'
  }
  const uint64_t x = w ^ 0x3030303030303030ull;   // a digit is 0 .. 9 now
  const bool digits = ((((x & 0x7F7F7F7F7F7F7F7Full) + 0x7676767676767676ull) | x) & 0x8080808080808080ull) == 0;
  return digits && ndot <= 1 && l > ndot;
}
__device__ void check_scalar(uint32_t kind, const uint8_t* p, uint32_t l, uint32_t* err) {
  switch (kind) {
    case CK_INT: { int32_t v; if (!swar_plain_number(p, l, false) && parse_i32_text(p, l, &v)) set_err(err, VERR_BAD_INT); break; }
    case CK_FLOAT: if (!swar_plain_number(p, l, true) && !f32_text_ok(p, l)) set_err(err, VERR_BAD_FLOAT); break;
    case CK_STR: {   // only an escape can make a string value an error
      bool esc = false;
      uint32_t k = 0;
      for (; k + 8 <= l && !esc; k += 8) esc = eq_mask8(((const dl_u64*)(p + k))->v, 0x2525252525252525ull) != 0;
      for (; k < l; k++) esc |= p[k] == '%';
      if (esc) pct_decoded_len(p, l, &esc, err);
      break;
    }
    case CK_CHAR: {   // one character (the text is UTF-8: one byte that is not a continuation byte)
      uint32_t nc = 0;
      for (uint32_t k = 0; k < l; k++) nc += (p[k] & 0xC0u) != 0x80u;
      if (nc != 1) set_err(err, VERR_BAD_CHAR);
      break;
    }
    default: break;
  }
}
// the value [p, p+l) of a key nothing is extracted for (not "." -- the caller has looked); kind from type_lookup
__device__ __noinline__ void check_value(uint32_t kind, const uint8_t* p, uint32_t l, uint32_t* err) {
  if (kind == CK_NONE) return;
  if (kind == CK_FLAG) { set_err(err, VERR_INVALID_FLAG); return; }
  if ((kind & 7u) == CK_UNSUPPORTED) { set_err(err, VERR_UNSUPPORTED_INFO); return; }
  if (kind == CK_GT) { if (!gt_text_ok(p, l)) set_err(err, VERR_BAD_GT); return; }
  if (!(kind & CK_LIST)) { check_scalar(kind, p, l, err); return; }
  uint32_t a = 0;
  for (uint32_t k = 0; k <= l; k++) {
    if (k == l || p[k] == ',') {
      if (!(k - a == 1 && p[a] == '.')) check_scalar(kind & 7u, p + a, k - a, err);
      a = k + 1;
    }
  }
}
// The entries of an INFO field u[a, b) in file order: f(q, eq, e) for the entry [q, e) whose first '=' is at eq (~0 = a bare key);
// f returns true to stop.  The delimiters are taken eight bytes at a time.
template <typename F>
__device__ __forceinline__ void info_entries(const uint8_t* __restrict__ u, uint64_t a, uint64_t b, F f) {
  uint64_t q = a, eqp = ~0ull;
  for (uint64_t w0 = a; w0 < b; w0 += 8) {
    const uint64_t w = ((const dl_u64*)(u + w0))->v;  // reads past b stay inside the buffer slack
    uint64_t m = eq_mask8(w, 0x3B3B3B3B3B3B3B3Bull) | eq_mask8(w, 0x3D3D3D3D3D3D3D3Dull);
    while (m) {
      const int bit = __builtin_ctzll(m);
      m &= m - 1;
      const uint64_t pq = w0 + (uint64_t)(bit >> 3);
      if (pq >= b) break;
      if (((w >> (bit - 7)) & 0xFF) == ';') { if (f(q, eqp, pq)) return; q = pq + 1; eqp = ~0ull; }
      else if (eqp == ~0ull) eqp = pq;
    }
  }
  if (q < b) f(q, eqp, b);
}

// ---- line keys ----------------------------------------------------------------------------------------------
// pos = POS; vend = noodles variant_end (INFO END, else POS + len(REF) - 1); flags bit0 = single-base ACGT SNV
// (get_variant_end's fast path, physical_exec.rs:646-667), bit1 = blank line, bit2 = POS needs more than 32 bits.
__global__ __launch_bounds__(256) void k_vcf_keys(const uint8_t* __restrict__ u, VcfLines L, uint32_t* __restrict__ pos,
                                                   uint32_t* __restrict__ vend, uint8_t* __restrict__ flags, int need_end,
                                                   VcfTypeTable T, uint32_t* __restrict__ err) {
  __shared__ VcfTypeSlot s_slots[VCF_TT_LDS_SLOTS];
  if (need_end) T = stage_type_table(T, s_slots);
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= L.n_lines) return;
  uint64_t a, b;
  const uint64_t s = line_start(L, i);
  uint64_t e = line_end_raw(L, i);
  if (e > s && u[e - 1] == '\r') e--;
  if (e == s) { flags[i] = 2; pos[i] = 0; vend[i] = 0; set_err(err, VERR_BLANK_LINE); return; }
  if (line_te(L, i) - line_tb(L, i) < 7) { set_err(err, VERR_SHORT_RECORD); flags[i] = 2; return; }
  field_span(L, u, i, 1, &a, &b);
  uint32_t p = 0;
  bool wide = false, zero = false;
  if (parse_u32_pos(u + a, (uint32_t)(b - a), &p, &wide, &zero)) { set_err(err, VERR_BAD_POS); p = 0; }
  else if (zero) set_err(err, VERR_MISSING_START);
  pos[i] = p;
  uint64_t ra, rb, aa, ab;
  field_span(L, u, i, 3, &ra, &rb);
  field_span(L, u, i, 4, &aa, &ab);
  const uint32_t rl = (uint32_t)(rb - ra);
  uint8_t fl = 0;
  if (rl == 1 && ab - aa == 1) {
    const uint8_t r = u[ra], al = u[aa];
    const bool rok = r == 'A' || r == 'C' || r == 'G' || r == 'T';
    const bool aok = al == 'A' || al == 'C' || al == 'G' || al == 'T';
    if (rok && aok) fl = 1;
  }
  if (wide) fl |= 4;   // bit2: POS >= 2^32 (pos / vend hold the wrapped values the reference's columns show)
  flags[i] = fl;
  uint32_t ve = p + rl - 1;
  // noodles' variant_end = `info.get(header, "END")`: the entries are walked and typed one after the other up to the first END;
  // get_variant_end does not ask for it when the record is a single-base ACGT substitution (physical_exec.rs:646-667)
  // (need_end bit 8: an indexed scan -- noodles' query asks every record for its variant_end to test the overlap)
  if (need_end && (!(fl & 1) || (need_end & 0x100))) {
    uint64_t ia, ib;
    field_span(L, u, i, 7, &ia, &ib);
    if (!(ib - ia == 1 && u[ia] == '.')) {
      const uint32_t end_kind = ((uint32_t)need_end & 0xFFu) - 1u;
      info_entries(u, ia, ib, [&](uint64_t q, uint64_t eqp, uint64_t e) -> bool {
        if (e == q) return false;                       // empty entry (";;")
        if (eqp == ~0ull) return e - q == 3 && u[q] == 'E' && u[q + 1] == 'N' && u[q + 2] == 'D';   // bare key: nothing to type
        const uint32_t kl = (uint32_t)(eqp - q), vl = (uint32_t)(e - eqp - 1);
        const uint8_t* v = u + eqp + 1;
        const bool is_end = kl == 3 && u[q] == 'E' && u[q + 1] == 'N' && u[q + 2] == 'D';
        if (!(vl == 1 && v[0] == '.')) {
          if (is_end && end_kind == CK_INT) {
            int32_t ev;
            if (parse_i32_text(v, vl, &ev) || ev <= 0) set_err(err, VERR_BAD_END);
            else ve = (uint32_t)ev;
          } else check_value(is_end ? end_kind : type_lookup(T, u + q, kl), v, vl, err);
        }
        return is_end;
      });
    }
  }
  vend[i] = ve;
}
void launch_vcf_keys(const uint8_t* u, VcfLines L, uint32_t* pos, uint32_t* vend, uint8_t* flags, int need_end, VcfTypeTable T,
                     uint32_t* err, hipStream_t st) {
  if (!L.n_lines) return;
  hipLaunchKernelGGL(k_vcf_keys, dim3((uint32_t)((L.n_lines + 255) / 256)), dim3(256), 0, st, u, L, pos, vend, flags, need_end, T, err);
}

// ---- row selection ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool bytes_eq(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb) {
  if (la != lb) return false;
  for (uint32_t k = 0; k < la; k++) if (a[k] != b[k]) return false;
  return true;
}
__device__ bool eval_terms(const VcfFilterTerm* __restrict__ terms, int n_terms, const uint8_t* __restrict__ strs,
                           const uint8_t* chrom, uint32_t lchrom, const uint8_t* id, uint32_t lid, double start, double end) {
  for (int t = 0; t < n_terms; t++) {
    const VcfFilterTerm& T = terms[t];
    const bool is_str = T.field == 0 || T.field == 3;
    const uint8_t* sv = T.field == 0 ? chrom : id;
    const uint32_t sl = T.field == 0 ? lchrom : lid;
    const double nv = T.field == 1 ? start : end;
    bool ok = true;
    if (T.op <= BIOSCAN_OP_GE) {
      if (is_str) {
        const bool eq = bytes_eq(sv, sl, strs + T.str_off[0], T.str_len[0]);
        ok = T.op == BIOSCAN_OP_EQ ? eq : (T.op == BIOSCAN_OP_NE ? !eq : true);
      } else {
        const double lv = T.vals[0];
        switch (T.op) {
          case BIOSCAN_OP_EQ: ok = nv == lv; break;
          case BIOSCAN_OP_NE: ok = nv != lv; break;
          case BIOSCAN_OP_LT: ok = nv < lv; break;
          case BIOSCAN_OP_LE: ok = nv <= lv; break;
          case BIOSCAN_OP_GT: ok = nv > lv; break;
          default: ok = nv >= lv; break;
        }
      }
    } else if (T.op == BIOSCAN_OP_BETWEEN || T.op == BIOSCAN_OP_NOT_BETWEEN) {
      const bool bt = nv >= T.vals[0] && nv <= T.vals[1];
      ok = T.op == BIOSCAN_OP_BETWEEN ? bt : !bt;
    } else {
      // a list longer than eight literals spans consecutive terms (`more`): the verdict is taken over the whole list
      const bool neg = T.op == BIOSCAN_OP_NOT_IN;
      bool hit = false, has_null = false;
      for (;;) {
        const VcfFilterTerm& Gt = terms[t];
        for (int k = 0; k < Gt.n_vals && !hit; k++)
          hit = is_str ? bytes_eq(sv, sl, strs + Gt.str_off[k], Gt.str_len[k]) : nv == Gt.vals[k];
        has_null = has_null || Gt.has_null;
        if (!Gt.more || t + 1 >= n_terms) break;
        t++;
      }
      ok = hit ? !neg : (!has_null && neg);
    }
    if (!ok) return false;
  }
  return true;
}

__global__ __launch_bounds__(256) void k_vcf_row_flags(const uint8_t* __restrict__ u, VcfLines L, const uint32_t* __restrict__ pos,
                                                        const uint32_t* __restrict__ vend, const uint8_t* __restrict__ flags,
                                                        VcfRowSelect S, const uint64_t* __restrict__ chunks,
                                                        const VcfFilterTerm* __restrict__ terms, const uint8_t* __restrict__ strs,
                                                        uint32_t* __restrict__ keep) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t i = S.i_lo + k;
  if (i >= S.i_hi) return;
  uint32_t kp = 0;
  do {
    if (flags[i] & 2) break;
    const uint64_t s = line_start(L, i);
    if (S.mode == 1) {
      bool in = false;
      for (int c = 0; c < S.n_chunks && !in; c++) in = s >= chunks[2 * c] && s < chunks[2 * c + 1];
      if (!in) break;
    }
    uint64_t ca, cb;
    field_span(L, u, i, 0, &ca, &cb);
    const uint32_t p = pos[i];
    if (S.mode == 1) {
      if (!bytes_eq(u + ca, (uint32_t)(cb - ca), strs + S.chrom_off, S.chrom_len)) break;
      // noodles intersects() on the positions as parsed: a POS beyond 2^32 lies behind every bounded interval
      if (flags[i] & 4) { if (S.end1 > 0) break; }
      else if (!((int64_t)p <= S.q_end1 && (int64_t)vend[i] >= S.q_start1)) break;
      if (S.start1 > 0 && (int64_t)p < S.start1) break;                           // physical_exec.rs:2886-2895
      if (S.end1 > 0 && (int64_t)p > S.end1) break;
    }
    if (S.n_terms) {
      uint64_t ia, ib;
      field_span(L, u, i, 2, &ia, &ib);
      uint32_t lid = (uint32_t)(ib - ia);
      if (lid == 1 && u[ia] == '.') lid = 0;
      const uint32_t endcol = (flags[i] & 1) ? p : vend[i];
      const double st = (double)(S.zero_based ? p - 1 : p);
      if (!eval_terms(terms, S.n_terms, strs, u + ca, (uint32_t)(cb - ca), u + ia, lid, st, (double)endcol)) break;
    }
    kp = 1;
  } while (false);
  keep[k] = kp;
}
void launch_vcf_row_flags(const uint8_t* u, VcfLines L, const uint32_t* pos, const uint32_t* vend, const uint8_t* flags,
                          VcfRowSelect S, const uint64_t* chunks, const VcfFilterTerm* terms, const uint8_t* strs, uint32_t* keep,
                          hipStream_t st) {
  if (S.i_hi <= S.i_lo) return;
  const uint64_t n = S.i_hi - S.i_lo;
  hipLaunchKernelGGL(k_vcf_row_flags, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, L, pos, vend, flags, S, chunks, terms,
                     strs, keep);
}
__global__ __launch_bounds__(256) void k_vcf_compact(const uint32_t* __restrict__ keep, const uint64_t* __restrict__ scan, uint64_t n,
                                                      uint64_t i_lo, uint64_t* __restrict__ rows, uint64_t row_base, uint64_t cap) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n || !keep[k]) return;
  const uint64_t o = scan[k];
  if (o < cap) rows[row_base + o] = i_lo + k;
}
void launch_vcf_compact(const uint32_t* keep, const uint64_t* scan, uint64_t n, uint64_t i_lo, uint64_t* rows, uint64_t row_base,
                        uint64_t cap, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_vcf_compact, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, keep, scan, n, i_lo, rows, row_base, cap);
}
// first line whose start is >= off (lines are sorted): used to bound a region's line range
__global__ void k_vcf_line_lower_bound(VcfLines L, uint64_t off, unsigned long long* result) {
  uint64_t lo = 0, hi = L.n_lines;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (line_start(L, mid) < off) lo = mid + 1; else hi = mid;
  }
  result[0] = lo;
}
void launch_vcf_line_lower_bound(VcfLines L, uint64_t off, unsigned long long* result, hipStream_t st) {
  hipLaunchKernelGGL(k_vcf_line_lower_bound, dim3(1), dim3(1), 0, st, L, off, result);
}
__global__ __launch_bounds__(256) void k_iota_rows(uint64_t* rows, uint64_t n) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k < n) rows[k] = k;
}
void launch_vcf_iota_rows(uint64_t* rows, uint64_t n, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_iota_rows, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, rows, n);
}

// ---- core columns -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vcf_core(const uint8_t* __restrict__ u, VcfLines L, const uint64_t* __restrict__ rows,
                                                   uint64_t n, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ vend,
                                                   const uint8_t* __restrict__ flags, VcfCoreCols C, int zero_based,
                                                   uint32_t* __restrict__ err) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool act = r < n;
  const uint64_t i = act ? rows[r] : 0;
  uint64_t a = 0, b = 0;
  if (act) {
    if (C.src_chrom) { field_span(L, u, i, 0, &a, &b); C.src_chrom[r] = a; C.len_chrom[r] = (uint32_t)(b - a); }
    const uint32_t p = pos[i];
    if (C.start) C.start[r] = zero_based ? p - 1 : p;
    if (C.end) C.end[r] = (flags[i] & 1) ? p : vend[i];
    if (C.src_id) {
      field_span(L, u, i, 2, &a, &b);
      uint32_t l = (uint32_t)(b - a);
      if (l == 1 && u[a] == '.') l = 0;
      C.src_id[r] = a; C.len_id[r] = l;
    }
    if (C.src_ref) { field_span(L, u, i, 3, &a, &b); C.src_ref[r] = a; C.len_ref[r] = (uint32_t)(b - a); }
    if (C.src_alt) {
      field_span(L, u, i, 4, &a, &b);
      uint32_t l = (uint32_t)(b - a);
      if (l == 1 && u[a] == '.') l = 0;
      C.src_alt[r] = a; C.len_alt[r] = l;
    }
    if (C.src_filter) {
      field_span(L, u, i, 6, &a, &b);
      uint32_t l = (uint32_t)(b - a);
      if (l == 1 && u[a] == '.') l = 0;
      C.src_filter[r] = a; C.len_filter[r] = l;
    }
  }
  if (C.qual) {
    bool valid = false;
    if (act) {
      field_span(L, u, i, 5, &a, &b);
      double q = 0.0;
      if (!(b - a == 1 && u[a] == '.')) {
        float f;
        const int rc = parse_f32_text(u + a, (uint32_t)(b - a), &f);
        if (rc == 1) set_err(err, VERR_BAD_QUAL);
        else { q = (double)f; valid = true; if (rc == 2) f32_defer(err, u + a, (uint32_t)(b - a), &C.qual[r], 1, f); }
      }
      C.qual[r] = q;
    }
    const unsigned long long m = __ballot(valid);
    if ((threadIdx.x & 63) == 0 && (r & ~63ull) < n) C.v_qual[r >> 6] = m;
  }
}
void launch_vcf_core(const uint8_t* u, VcfLines L, const uint64_t* rows, uint64_t n, const uint32_t* pos, const uint32_t* vend,
                     const uint8_t* flags, VcfCoreCols C, int zero_based, uint32_t* err, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_vcf_core, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, L, rows, n, pos, vend, flags, C, zero_based, err);
}
__global__ __launch_bounds__(256) void k_replace_byte(uint8_t* __restrict__ d, uint64_t n, uint8_t from, uint8_t to) {
  const uint64_t k = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  for (uint64_t j = k; j < n && j < k + 16; j++) if (d[j] == from) d[j] = to;
}
void launch_replace_byte(uint8_t* d, uint64_t n, uint8_t from, uint8_t to, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_replace_byte, dim3((uint32_t)((n + 4095) / 4096)), dim3(256), 0, st, d, n, from, to);
}

// ---- INFO ---------------------------------------------------------------------------------------------------
// One pass over the INFO field of each row: for every selected key k, sp_off/sp_len/sp_state[k*n + r]
// (state 0 absent, 1 `key=value`, 2 bare key).  A selected key that occurs twice is an error (the reference
// would append twice and misalign the column).
__global__ __launch_bounds__(256) void k_vcf_info_locate(const uint8_t* __restrict__ u, VcfLines L, const uint64_t* __restrict__ rows,
                                                          uint64_t n, const uint8_t* __restrict__ keys, const uint32_t* __restrict__ key_off,
                                                          const uint8_t* __restrict__ key_unsupported, int K, VcfTypeTable T,
                                                          uint64_t* __restrict__ sp_off, uint32_t* __restrict__ sp_len,
                                                          uint8_t* __restrict__ sp_state, uint32_t* __restrict__ err) {
  __shared__ VcfTypeSlot s_slots[VCF_TT_LDS_SLOTS];
  T = stage_type_table(T, s_slots);
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const uint64_t i = rows[r];
  for (int k = 0; k < K; k++) sp_state[(uint64_t)k * n + r] = 0;
  uint64_t a, b;
  field_span(L, u, i, 7, &a, &b);
  if (b - a == 1 && u[a] == '.') return;
  info_entries(u, a, b, [&](uint64_t q, uint64_t eqp, uint64_t ve) -> bool {
    if (ve == q) return false;   // empty entry
    const bool has_val = eqp != ~0ull;
    const uint64_t ke = has_val ? eqp : ve;
    const uint32_t kl = (uint32_t)(ke - q);
    const uint32_t vl = has_val ? (uint32_t)(ve - ke - 1) : 0u;
    const bool value = has_val && !(vl == 1 && u[ke + 1] == '.');
    auto take = [&](int k) {
      const uint64_t o = (uint64_t)k * n + r;
      if (sp_state[o]) set_err(err, VERR_DUP_INFO_KEY);
      if (value && key_unsupported[k]) set_err(err, VERR_UNSUPPORTED_INFO);
      sp_state[o] = has_val ? 1 : 2;
      sp_off[o] = has_val ? ke + 1 : ke;
      sp_len[o] = vl;
    };
    uint32_t kind = 0, sel = 0;
    if (T.has_sel) {
      // one lookup says both: which selected key this is, or how the header types a key nothing is extracted for
      kind = type_lookup(T, u + q, kl, &sel);
      if (sel) { take((int)sel - 1); return false; }
    } else {
      for (int k = 0; kl && k < K; k++) {
        const uint32_t ko = key_off[k], kn = key_off[k + 1] - ko;
        if (kn == kl && bytes_eq(u + q, kl, keys + ko, kn)) { take(k); return false; }
      }
      kind = type_lookup(T, u + q, kl);
    }
    // no column for this key: typed by the header all the same (see VcfCheckKind)
    if (value) check_value(kind, u + ke + 1, vl, err);
    return false;
  });
}
void launch_vcf_info_locate(const uint8_t* u, VcfLines L, const uint64_t* rows, uint64_t n, const uint8_t* keys, const uint32_t* key_off,
                            const uint8_t* key_unsupported, int K, VcfTypeTable T, uint64_t* sp_off, uint32_t* sp_len,
                            uint8_t* sp_state, uint32_t* err, hipStream_t st) {
  if (!n || !K) return;
  hipLaunchKernelGGL(k_vcf_info_locate, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, L, rows, n, keys, key_off,
                     key_unsupported, K, T, sp_off, sp_len, sp_state, err);
}

// ---- typed span kernels (shared by INFO and FORMAT cells) -----------------------------------------------------
// kind: 0 Int32, 1 Float32.  A span holding "." is NULL; state 0 / 2 (absent / bare non-flag key) is NULL.
__global__ __launch_bounds__(256) void k_span_num(const uint8_t* __restrict__ u, const uint64_t* __restrict__ off,
                                                   const uint32_t* __restrict__ len, const uint8_t* __restrict__ state, uint64_t N,
                                                   int kind, uint32_t* __restrict__ values, uint64_t* __restrict__ valid,
                                                   uint32_t* __restrict__ err) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool v = false;
  if (c < N) {
    uint32_t out = 0;
    if (state[c] == 1) {
      const uint8_t* p = u + off[c];
      const uint32_t l = len[c];
      if (!(l == 1 && p[0] == '.')) {
        if (kind == 0) {
          int32_t x;
          if (parse_i32_text(p, l, &x)) set_err(err, VERR_BAD_INT); else { out = (uint32_t)x; v = true; }
        } else {
          float f;
          const int rc = parse_f32_text(p, l, &f);
          if (rc == 1) set_err(err, VERR_BAD_FLOAT); else { out = __float_as_uint(f); v = true; if (rc == 2) f32_defer(err, p, l, &values[c], 0, f); }
        }
      }
    }
    values[c] = out;
  }
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0 && (c & ~63ull) < N) valid[c >> 6] = m;
}
void launch_span_num(const uint8_t* u, const uint64_t* off, const uint32_t* len, const uint8_t* state, uint64_t N, int kind,
                     uint32_t* values, uint64_t* valid, uint32_t* err, hipStream_t st) {
  if (!N) return;
  hipLaunchKernelGGL(k_span_num, dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, st, u, off, len, state, N, kind, values, valid, err);
}
// Flag: present (bare) -> true; absent -> false; `key=.` -> false; any other value is an error
__global__ __launch_bounds__(256) void k_span_flag(const uint8_t* __restrict__ u, const uint64_t* __restrict__ off,
                                                    const uint32_t* __restrict__ len, const uint8_t* __restrict__ state, uint64_t N,
                                                    uint64_t* __restrict__ bits, uint32_t* __restrict__ err) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool v = false;
  if (c < N) {
    if (state[c] == 2) v = true;
    else if (state[c] == 1 && !(len[c] == 1 && u[off[c]] == '.')) set_err(err, VERR_INVALID_FLAG);
  }
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0 && (c & ~63ull) < N) bits[c >> 6] = m;
}
void launch_span_flag(const uint8_t* u, const uint64_t* off, const uint32_t* len, const uint8_t* state, uint64_t N, uint64_t* bits,
                      uint32_t* err, hipStream_t st) {
  if (!N) return;
  hipLaunchKernelGGL(k_span_flag, dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, st, u, off, len, state, N, bits, err);
}
// Utf8: out_len = span length (0 when NULL), validity; '%' (percent-encoding) is rejected, not decoded
__global__ __launch_bounds__(256) void k_span_str(const uint8_t* __restrict__ u, const uint64_t* __restrict__ off,
                                                   const uint32_t* __restrict__ len, const uint8_t* __restrict__ state, uint64_t N,
                                                   uint32_t* __restrict__ out_len, uint64_t* __restrict__ valid, uint32_t* __restrict__ pct_flag,
                                                   uint32_t* __restrict__ err) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool v = false;
  if (c < N) {
    uint32_t l = 0;
    if (state[c] == 1) {
      const uint8_t* p = u + off[c];
      l = len[c];
      if (l == 1 && p[0] == '.') l = 0;
      else {
        v = true;
        bool pct = false;
        l = pct_decoded_len(p, l, &pct, err);
        if (pct) atomicOr(pct_flag, 1u);
      }
    }
    out_len[c] = l;
  }
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0 && (c & ~63ull) < N) valid[c >> 6] = m;
}
void launch_span_str(const uint8_t* u, const uint64_t* off, const uint32_t* len, const uint8_t* state, uint64_t N, uint32_t* out_len,
                     uint64_t* valid, uint32_t* pct_flag, uint32_t* err, hipStream_t st) {
  if (!N) return;
  hipLaunchKernelGGL(k_span_str, dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, st, u, off, len, state, N, out_len, valid, pct_flag, err);
}
// copy with percent-decoding: row r produces off64[r+1] - off64[r] bytes from u[src[r] ..]
__global__ __launch_bounds__(256) void k_scatter_pct(const uint8_t* __restrict__ u, const uint64_t* __restrict__ src, uint64_t n,
                                                      const uint64_t* __restrict__ off64, uint8_t* __restrict__ dst) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const uint64_t o = off64[r];
  const uint32_t L = (uint32_t)(off64[r + 1] - o);
  const uint8_t* p = u + src[r];
  for (uint32_t k = 0; k < L; k++) {
    uint32_t c = *p;
    if (c == '%') {
      const int h = hexval(p[1]), lo = h >= 0 ? hexval(p[2]) : -1;
      if (lo >= 0) { c = (uint32_t)(h * 16 + lo); p += 2; }
    }
    dst[o + k] = (uint8_t)c;
    p++;
  }
}
void launch_scatter_pct(const uint8_t* u, const uint64_t* src, uint64_t n, const uint64_t* off64, uint8_t* dst, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_scatter_pct, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, src, n, off64, dst);
}
// list element count: "." / absent -> NULL list (0 elements); else commas + 1
__global__ __launch_bounds__(256) void k_span_list_count(const uint8_t* __restrict__ u, const uint64_t* __restrict__ off,
                                                          const uint32_t* __restrict__ len, const uint8_t* __restrict__ state, uint64_t N,
                                                          uint32_t* __restrict__ cnt, uint64_t* __restrict__ valid) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool v = false;
  if (c < N) {
    uint32_t k = 0;
    if (state[c] == 1) {
      const uint8_t* p = u + off[c];
      const uint32_t l = len[c];
      if (!(l == 1 && p[0] == '.')) {
        v = true;
        k = 1;
        for (uint32_t j = 0; j < l; j++) k += p[j] == ',';
      }
    }
    cnt[c] = k;
  }
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0 && (c & ~63ull) < N) valid[c >> 6] = m;
}
void launch_span_list_count(const uint8_t* u, const uint64_t* off, const uint32_t* len, const uint8_t* state, uint64_t N, uint32_t* cnt,
                            uint64_t* valid, hipStream_t st) {
  if (!N) return;
  hipLaunchKernelGGL(k_span_list_count, dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, st, u, off, len, state, N, cnt, valid);
}
// list elements: kind 0 Int32, 1 Float32 -> values + per-element validity bytes; kind 2 Utf8 -> element
// source offsets / lengths + validity bytes.  eoff = exclusive scan of the counts.
__global__ __launch_bounds__(256) void k_span_list_elems(const uint8_t* __restrict__ u, const uint64_t* __restrict__ off,
                                                          const uint32_t* __restrict__ len, const uint8_t* __restrict__ state, uint64_t N,
                                                          const uint64_t* __restrict__ eoff, int kind, uint32_t* __restrict__ values,
                                                          uint64_t* __restrict__ esrc, uint32_t* __restrict__ elen,
                                                          uint8_t* __restrict__ evalid, uint32_t* __restrict__ pct_flag,
                                                          uint32_t* __restrict__ err) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  uint64_t o = eoff[c];
  const uint64_t oe = eoff[c + 1];
  if (o == oe) return;
  const uint64_t base = off[c];
  const uint8_t* p = u + base;
  const uint32_t l = len[c];
  uint32_t a = 0;
  for (uint32_t j = 0; j <= l; j++) {
    if (j == l || p[j] == ',') {
      const uint32_t el = j - a;
      const bool miss = el == 1 && p[a] == '.';
      if (kind == 2) {
        esrc[o] = base + a;
        uint32_t dl = 0;
        if (!miss) {
          bool pct = false;
          dl = pct_decoded_len(p + a, el, &pct, err);
          if (pct) atomicOr(pct_flag, 1u);
        }
        elen[o] = dl;
      } else {
        uint32_t out = 0;
        if (!miss) {
          if (kind == 0) {
            int32_t x;
            if (parse_i32_text(p + a, el, &x)) set_err(err, VERR_BAD_INT); else out = (uint32_t)x;
          } else {
            float f;
            const int rc = parse_f32_text(p + a, el, &f);
            if (rc == 1) set_err(err, VERR_BAD_FLOAT); else { out = __float_as_uint(f); if (rc == 2) f32_defer(err, p + a, el, &values[o], 0, f); }
          }
        }
        values[o] = out;
      }
      evalid[o] = miss ? 0 : 1;
      o++;
      a = j + 1;
    }
  }
}
void launch_span_list_elems(const uint8_t* u, const uint64_t* off, const uint32_t* len, const uint8_t* state, uint64_t N,
                            const uint64_t* eoff, int kind, uint32_t* values, uint64_t* esrc, uint32_t* elen, uint8_t* evalid,
                            uint32_t* pct_flag, uint32_t* err, hipStream_t st) {
  if (!N) return;
  hipLaunchKernelGGL(k_span_list_elems, dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, st, u, off, len, state, N, eoff, kind, values,
                     esrc, elen, evalid, pct_flag, err);
}
__global__ __launch_bounds__(256) void k_pack_bits(const uint8_t* __restrict__ bytes, uint64_t n, uint64_t* __restrict__ words) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const bool v = c < n && bytes[c];
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0 && (c & ~63ull) < n) words[c >> 6] = m;
}
void launch_pack_bits(const uint8_t* bytes, uint64_t n, uint64_t* words, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_pack_bits, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, bytes, n, words);
}
__global__ __launch_bounds__(256) void k_stride_offsets(uint64_t* off, uint64_t n, uint64_t stride) {
  const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (k <= n) off[k] = k * stride;
}
__global__ __launch_bounds__(256) void k_off64_to_32(const uint64_t* __restrict__ off, uint64_t n, int32_t* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (int32_t)off[i];
}
void launch_off64_to_32(const uint64_t* off, uint64_t n, int32_t* out, hipStream_t st) {
  if (n) hipLaunchKernelGGL(k_off64_to_32, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, off, n, out);
}
void launch_stride_offsets(uint64_t* off, uint64_t n, uint64_t stride, hipStream_t st) {
  hipLaunchKernelGGL(k_stride_offsets, dim3((uint32_t)((n + 256) / 256)), dim3(256), 0, st, off, n, stride);
}

// ---- FORMAT -------------------------------------------------------------------------------------------------
// fpos[r*S + s] = index of selected FORMAT key s among the ':'-separated keys of row r's FORMAT column (-1 absent)
__global__ __launch_bounds__(256) void k_vcf_format_keys(const uint8_t* __restrict__ u, VcfLines L, const uint64_t* __restrict__ rows,
                                                          uint64_t n, const uint8_t* __restrict__ keys, const uint32_t* __restrict__ key_off,
                                                          int S, VcfTypeTable T, uint32_t char_mask, int16_t* __restrict__ fpos,
                                                          uint64_t* __restrict__ cmap, uint32_t* __restrict__ err) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  for (int s = 0; s < S; s++) fpos[r * S + s] = -1;
  cmap[r] = 0;
  uint64_t a, b;
  if (!field_span(L, u, rows[r], 8, &a, &b)) return;
  if (b == a || (b - a == 1 && u[a] == '.')) return;
  int j = 0;
  uint64_t q = a, cm = 0;
  while (q <= b) {
    uint64_t e = q;
    while (e < b && u[e] != ':') e++;
    const uint32_t kl = (uint32_t)(e - q);
    uint32_t ck = ~0u;
    for (int s = 0; s < S; s++) {
      const uint32_t ko = key_off[s], kn = key_off[s + 1] - ko;
      if (kn == kl && bytes_eq(u + q, kl, keys + ko, kn)) {
        // extracted (and typed) by the cell kernel; a selected Character scalar is a string there and still has to be one character
        ck = (s < 32 && ((char_mask >> s) & 1u)) ? (uint32_t)CK_CHAR : (uint32_t)CK_NONE;
        if (fpos[r * S + s] < 0) fpos[r * S + s] = (int16_t)j;
      }
    }
    if (j < 15) {
      if (ck == ~0u) ck = type_lookup(T, u + q, kl) & 15u;
      cm |= (uint64_t)ck << (4 * j);
    } else cm |= 15ull << 60;   // more keys than the map holds: k_vcf_format_check looks those up itself
    j++;
    q = e + 1;
  }
  cmap[r] = cm;
  if (cm) err[3] = 1u;   // some value of this chunk is checked without being extracted: k_vcf_format_check has to run
}
// check kind of the row's FORMAT key j (for the rows with more keys than a cmap holds; a selected key is then checked a
// second time, with the same outcome as its extraction)
__device__ __noinline__ uint32_t format_key_kind(const uint8_t* __restrict__ u, const VcfLines& L, uint64_t line, int j, const VcfTypeTable& T) {
  uint64_t a, b;
  if (!field_span(L, u, line, 8, &a, &b)) return CK_NONE;
  uint64_t q = a;
  for (int k = 0; q <= b; k++) {
    uint64_t e = q;
    while (e < b && u[e] != ':') e++;
    if (k == j) return type_lookup(T, u + q, (uint32_t)(e - q)) & 15u;
    q = e + 1;
  }
  return CK_NONE;
}
void launch_vcf_format_keys(const uint8_t* u, VcfLines L, const uint64_t* rows, uint64_t n, const uint8_t* keys, const uint32_t* key_off,
                            int S, VcfTypeTable T, uint32_t char_mask, int16_t* fpos, uint64_t* cmap, uint32_t* err, hipStream_t st) {
  if (!n || !S) return;
  hipLaunchKernelGGL(k_vcf_format_keys, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, u, L, rows, n, keys, key_off, S, T,
                     char_mask, fpos, cmap, err);
}
// The values of FORMAT keys the cell kernel does not extract, typed by the header as `sample.iter(header)` does for every value
// of a selected sample (see VcfCheckKind).  One thread per (row, selected sample); run only when k_vcf_format_keys has found
// such a key in the chunk (err[3]) -- a scan of all the table's FORMAT fields over a file that declares its keys never does.
__global__ __launch_bounds__(256) void k_vcf_format_check(const uint8_t* __restrict__ u, VcfLines L, const uint64_t* __restrict__ rows,
                                                           uint64_t n, const int32_t* __restrict__ sample_col, int NS,
                                                           const uint64_t* __restrict__ cmap, VcfTypeTable T, uint32_t* __restrict__ err) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= n * (uint64_t)NS) return;
  const uint64_t r = c / (uint64_t)NS;
  const uint64_t cm = cmap[r];
  if (cm == 0) return;
  uint64_t a, b;
  if (!field_span(L, u, rows[r], 9 + (uint32_t)sample_col[c - r * NS], &a, &b)) return;
  if (b == a || (b - a == 1 && u[a] == '.')) return;
  int j = 0;
  for (uint64_t q = a; q <= b; j++) {
    uint64_t e = q;
    while (e < b && u[e] != ':') e++;
    const uint32_t ck = j < 15 ? (uint32_t)((cm >> (4 * j)) & 15ull) : format_key_kind(u, L, rows[r], j, T);
    if (ck != CK_NONE && !(e - q == 1 && u[q] == '.')) check_value(ck, u + q, (uint32_t)(e - q), err);
    q = e + 1;
  }
}
void launch_vcf_format_check(const uint8_t* u, VcfLines L, const uint64_t* rows, uint64_t n, const int32_t* sample_col, int NS,
                             const uint64_t* cmap, VcfTypeTable T, uint32_t* err, hipStream_t st) {
  const uint64_t N = n * (uint64_t)NS;
  if (!N) return;
  hipLaunchKernelGGL(k_vcf_format_check, dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, st, u, L, rows, n, sample_col, NS, cmap, T, err);
}
// zeros a finished GT allele token loses when it is rendered again as an integer: all its leading zeros, but one digit stays
__device__ __forceinline__ uint32_t gt_strip_add(bool isdot, bool still_leading, uint32_t nz, uint32_t tl) {
  if (isdot || tl == 0) return 0u;
  return still_leading ? tl - 1u : nz;
}
// One thread per (row, selected sample) cell c = r*NS + os: the ':'-separated value of each selected FORMAT
// key -> sp_*[s*N + c] (N = n*NS).  gt_field = index of GT among the selected keys (-1 none): its span drops
// a leading phasing character and is validated (alleles are digits or '.', separators '/' or '|').
__global__ __launch_bounds__(256, 6) void k_vcf_format_cells(const uint8_t* __restrict__ u, VcfLines L, const uint64_t* __restrict__ rows,
                                                           uint64_t n, const int32_t* __restrict__ sample_col, int NS,
                                                           const int16_t* __restrict__ fpos, int S, int gt_field, VcfCellDirect D,
                                                           uint64_t* __restrict__ sp_off, uint32_t* __restrict__ sp_len,
                                                           uint8_t* __restrict__ sp_state, uint32_t* __restrict__ err) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t N = n * (uint64_t)NS;
  const bool act = c < N;
  // 32-bit division when the cell index fits (64-bit division is emulated with ~100 instructions)
  const uint64_t r = !act ? 0 : (N <= 0xFFFFFFFFull ? (uint64_t)((uint32_t)c / (uint32_t)NS) : c / NS);
  const int os = (int)(c - r * NS);
  uint32_t dfix = 0;               // bit s: the float of direct field s is to be rounded exactly afterwards
  uint32_t dval[VCF_MAX_DIRECT];   // direct Int32 / Float32 value, or the length of a direct string
  uint32_t dsrc[VCF_MAX_DIRECT];   // direct string: start relative to the cell
  uint32_t dok = 0;                // bit s: direct field s has a value
#pragma unroll
  for (int s = 0; s < VCF_MAX_DIRECT; s++) { dval[s] = 0; dsrc[s] = 0; }
  uint64_t a = 0, b = 0;
  if (act) {
    for (int s = 0; s < S; s++) if (s >= VCF_MAX_DIRECT || D.kind[s] == 0) sp_state[(uint64_t)s * N + c] = 0;
    bool have = field_span(L, u, rows[r], 9 + (uint32_t)sample_col[os], &a, &b);
    const uint32_t len = have ? (uint32_t)(b - a) : 0;
    // the first 16 bytes of the cell travel in registers (one unaligned 16-byte load; the text buffer has slack)
    uint64_t w_lo = 0, w_hi = 0;
    if (len) { w_lo = ((const dl_u64*)(u + a))->v; w_hi = ((const dl_u64*)(u + a + 8))->v; }
    auto byte_at = [&](uint32_t k) -> uint32_t {
      if (k < 8) return (uint32_t)(w_lo >> (8 * k)) & 0xFFu;
      if (k < 16) return (uint32_t)(w_hi >> (8 * (k - 8))) & 0xFFu;
      return u[a + k];
    };
    if (len == 1 && byte_at(0) == '.') have = false;
    // one forward pass: sub-field j belongs to selected key `cur` (-1: not selected).  The row's key positions are
    // fetched once (independent loads) and packed into a nibble map, so the byte loop touches no memory.
    uint64_t jmap = 0;   // nibble j = selected key index + 1
    uint32_t kmap = 0;   // 2 bits per selected key: its direct kind
#pragma unroll
    for (int s = 0; s < VCF_MAX_DIRECT; s++) {
      if (s < S) {
        const int fp = fpos[r * S + s];
        if (fp >= 0 && fp < 16 && ((jmap >> (4 * fp)) & 15ull) == 0) jmap |= (uint64_t)(s + 1) << (4 * fp);
        kmap |= (uint32_t)D.kind[s] << (2 * s);
      }
    }
    int j = 0, cur = -1, ckind = 0;
    auto select = [&]() {
      cur = -1; ckind = 0;
      if (j < 16) cur = (int)((jmap >> (4 * j)) & 15ull) - 1;
      if (S > VCF_MAX_DIRECT || j >= 16) {   // rare: more selected keys / sub-fields than the map holds
        for (int s = 0; s < S; s++) if (fpos[r * S + s] == j) { cur = s; break; }
      }
      if (cur >= 0 && cur < VCF_MAX_DIRECT) ckind = (int)((kmap >> (2 * cur)) & 3u);
    };
    // Fast path (cell of at most 16 bytes -- it sits in the two registers -- and the nibble maps hold every key): the
    // ':' positions come from one SWAR test, then the cell is walked sub-field by sub-field, each with a tight loop
    // for its kind.  All lanes of a wave are at the same sub-field together (a row's FORMAT is shared by its
    // samples), which the byte-by-byte state machine below cannot offer: there a lane at a ':' and a lane inside a
    // value make the wave execute both branches at every step.
    const bool fast = have && len != 0 && len <= 16 && S <= VCF_MAX_DIRECT;
    if (fast) {
      const uint64_t c8 = 0x3A3A3A3A3A3A3A3Aull;
      auto movemask = [](uint64_t m) -> uint32_t { return (uint32_t)((((m >> 7) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56); };
      uint32_t cm = movemask(eq_mask8(w_lo, c8)) | (movemask(eq_mask8(w_hi, c8)) << 8);
      cm &= len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u);
      uint32_t start = 0;
      for (;;) {
        const uint32_t e = cm ? (uint32_t)__builtin_ctz(cm) : len;
        cm &= cm - 1u;
        const uint32_t sl = e - start;
        select();
        const bool missing = sl == 1 && byte_at(start) == '.';
        if (cur >= 0 && !missing) {
          if (ckind == 1) {           // Int32: [+-]?digits
            int64_t acc = 0;
            uint32_t ndig = 0;
            bool neg = false, bad = false;
            for (uint32_t k = start; k < e; k++) {
              const uint32_t ch = byte_at(k);
              const uint32_t dgt = ch - '0';
              if (dgt <= 9) { if (acc < 100000000000ll) acc = acc * 10 + dgt; ndig++; }
              else if ((ch == '-' || ch == '+') && k == start) neg = ch == '-';
              else bad = true;
            }
            if (bad || ndig == 0 || (neg ? acc > 2147483648ll : acc > 2147483647ll)) set_err(err, VERR_BAD_INT);
            else {
              const uint32_t v = (uint32_t)(int32_t)(neg ? -acc : acc);
#pragma unroll
              for (int s = 0; s < VCF_MAX_DIRECT; s++) if (s == cur) { dval[s] = v; dok |= 1u << s; }
            }
          } else if (ckind == 2) {    // Float32: correctly rounded parse of the span
            float f;
            const int rc = parse_f32_text(u + a + start, sl, &f);
            if (rc == 1) set_err(err, VERR_BAD_FLOAT);
            else {
#pragma unroll
              for (int s = 0; s < VCF_MAX_DIRECT; s++) if (s == cur) { dval[s] = __float_as_uint(f); dok |= 1u << s; }
              if (rc == 2) {  // rounding not proven: queued for k_f32_fix where the cell is stored (its span rides in dsrc)
                if (start >= (1u << 20) || sl >= (1u << 12)) set_err(err, VERR_FLOAT_PRECISION);
                else {
#pragma unroll
                  for (int s = 0; s < VCF_MAX_DIRECT; s++) if (s == cur) { dsrc[s] = start | (sl << 20); dfix |= 1u << s; }
                }
              }
            }
          } else if (ckind == 3) {    // GT as a direct string: validated, leading phasing mark dropped
            // (noodles parses an allele as an integer and the reference renders it again: "01/1" comes out as "1/1" --
            // `strip` counts the leading zeros that go; k_gt_render writes such a cell, see gt_strip_add)
            uint32_t tl = 0, nz = 0, strip = 0;
            bool isdot = false, lead0 = false, gt_ok = true;
            for (uint32_t k = start; k < e; k++) {
              const uint32_t ch = byte_at(k);
              if (ch == '/' || ch == '|') {
                if (k != start) { if (tl == 0) gt_ok = false; strip += gt_strip_add(isdot, lead0, nz, tl); tl = 0; isdot = false; lead0 = false; nz = 0; }
              } else if (ch == '.') { if (tl) gt_ok = false; isdot = true; tl = 1; }
              else if (ch >= '0' && ch <= '9') {
                if (isdot) gt_ok = false;
                if (tl == 0) { lead0 = ch == '0'; nz = lead0 ? 1u : 0u; } else if (lead0) { if (ch == '0') nz++; else lead0 = false; }
                tl++;
              } else gt_ok = false;
            }
            strip += gt_strip_add(isdot, lead0, nz, tl);
            uint32_t x = start;
            const uint32_t c0 = byte_at(start);
            if (c0 == '/' || c0 == '|') x++;
            if (!gt_ok || tl == 0 || x >= e) set_err(err, VERR_BAD_GT);
            if (strip) err[2] = 1u;
#pragma unroll
            for (int s = 0; s < VCF_MAX_DIRECT; s++) if (s == cur) { dval[s] = e - x - strip; dsrc[s] = x; dok |= 1u << s; }
          } else {                    // span for the typed span kernels (strings, lists)
            const uint64_t o = (uint64_t)cur * N + c;
            sp_state[o] = 1;
            sp_off[o] = a + start;
            sp_len[o] = sl;
          }
        }
        j++;
        start = e + 1;
        if (e >= len) break;
      }
    }
    if (have && len && !fast) select();
    uint32_t start = 0, ndig = 0, tl = 0, nz = 0, strip = 0;
    int64_t acc = 0;
    bool neg = false, bad = false, isdot = false, lead0 = false, gt_ok = true;
    for (uint32_t k = 0; have && !fast && k <= len; k++) {
      const uint32_t ch = k < len ? byte_at(k) : (uint32_t)':';
      if (ch == ':') {
        const uint32_t sl = k - start;
        const bool missing = sl == 1 && byte_at(start) == '.';
        if (cur >= 0 && !missing) {
          if (ckind == 1) {           // Int32: [+-]?digits
            if (bad || ndig == 0 || (neg ? acc > 2147483648ll : acc > 2147483647ll)) set_err(err, VERR_BAD_INT);
            else {
              const uint32_t v = (uint32_t)(int32_t)(neg ? -acc : acc);
#pragma unroll
              for (int s = 0; s < VCF_MAX_DIRECT; s++) if (s == cur) { dval[s] = v; dok |= 1u << s; }
            }
          } else if (ckind == 2) {    // Float32: correctly rounded parse of the span
            float f;
            const int rc = parse_f32_text(u + a + start, sl, &f);
            if (rc == 1) set_err(err, VERR_BAD_FLOAT);
            else {
#pragma unroll
              for (int s = 0; s < VCF_MAX_DIRECT; s++) if (s == cur) { dval[s] = __float_as_uint(f); dok |= 1u << s; }
              if (rc == 2) {  // rounding not proven: queued for k_f32_fix where the cell is stored (its span rides in dsrc)
                if (start >= (1u << 20) || sl >= (1u << 12)) set_err(err, VERR_FLOAT_PRECISION);
                else {
#pragma unroll
                  for (int s = 0; s < VCF_MAX_DIRECT; s++) if (s == cur) { dsrc[s] = start | (sl << 20); dfix |= 1u << s; }
                }
              }
            }
          } else if (ckind == 3) {    // GT as a direct string: validated on the fly, leading phasing mark dropped
            uint32_t x = start;
            const uint32_t c0 = byte_at(start);
            if (c0 == '/' || c0 == '|') x++;
            strip += gt_strip_add(isdot, lead0, nz, tl);
            if (!gt_ok || tl == 0 || x >= k) set_err(err, VERR_BAD_GT);
            if (strip) err[2] = 1u;
#pragma unroll
            for (int s = 0; s < VCF_MAX_DIRECT; s++) if (s == cur) { dval[s] = k - x - strip; dsrc[s] = x; dok |= 1u << s; }
          } else {                    // span for the typed span kernels (strings, lists)
            const uint64_t o = (uint64_t)cur * N + c;
            sp_state[o] = 1;
            sp_off[o] = a + start;
            sp_len[o] = sl;
          }
        }
        j++;
        start = k + 1;
        if (k < len) select();
        acc = 0; ndig = 0; neg = false; bad = false; tl = 0; isdot = false; lead0 = false; gt_ok = true; nz = 0; strip = 0;
      } else if (cur >= 0) {
        if (ckind == 1) {
          const uint32_t dgt = ch - '0';
          if (dgt <= 9) { if (acc < 100000000000ll) acc = acc * 10 + dgt; ndig++; }
          else if ((ch == '-' || ch == '+') && k == start) neg = ch == '-';
          else bad = true;
        } else if (ckind == 3) {
          if (ch == '/' || ch == '|') {
            if (k != start) { if (tl == 0) gt_ok = false; strip += gt_strip_add(isdot, lead0, nz, tl); tl = 0; isdot = false; lead0 = false; nz = 0; }
          } else if (ch == '.') { if (tl) gt_ok = false; isdot = true; tl = 1; }
          else if (ch >= '0' && ch <= '9') {
            if (isdot) gt_ok = false;
            if (tl == 0) { lead0 = ch == '0'; nz = lead0 ? 1u : 0u; } else if (lead0) { if (ch == '0') nz++; else lead0 = false; }
            tl++;
          }
          else gt_ok = false;
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < VCF_MAX_DIRECT; s++) {
    if (s < S && D.kind[s] != 0) {   // wave-uniform
      if (act) {
        D.values[s][c] = dval[s];
        if (D.kind[s] == 3) D.src[s][c] = a + dsrc[s];
        if ((dfix >> s) & 1u) f32_defer(err, u + a + (dsrc[s] & 0xFFFFFu), dsrc[s] >> 20, &D.values[s][c], 0, __uint_as_float(dval[s]));
      }
      const unsigned long long m = __ballot((dok >> s) & 1u);
      if ((threadIdx.x & 63) == 0 && (c & ~63ull) < N) D.valid[s][c >> 6] = m;
    }
  }
}
// GT cells whose alleles carry leading zeros: the value of the column is the re-rendered genotype (alleles as the integers
// noodles parsed, physical_exec.rs:1675-1694), which is not a substring of the text.  The cell kernel has sized the cell for
// the rendered form (and raised err[2]); this kernel, run only then, writes those cells.
__global__ __launch_bounds__(256) void k_gt_render(const uint8_t* __restrict__ u, const uint64_t* __restrict__ src, const uint64_t* __restrict__ off,
                                                    uint8_t* __restrict__ values, uint64_t N) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const uint64_t o0 = off[i], o1 = off[i + 1];
  if (o1 == o0) return;
  const uint8_t* p = u + src[i];
  // raw length of the genotype: up to the next ':' / tab / end of line
  uint32_t raw = 0;
  while (true) { const uint8_t c = p[raw]; if (c == ':' || c == '\t' || c == '\n' || c == '\r' || c == 0) break; raw++; }
  if ((uint64_t)raw == o1 - o0) return;   // nothing was stripped: the scatter copied the cell
  uint8_t* q = values + o0;
  uint32_t k = 0;
  while (k < raw && q < values + o1) {
    const uint8_t c = p[k];
    if (c >= '0' && c <= '9') {
      uint32_t e = k;
      while (e < raw && p[e] >= '0' && p[e] <= '9') e++;
      uint32_t b = k;
      while (b + 1 < e && p[b] == '0') b++;   // leading zeros go, one digit stays
      for (; b < e && q < values + o1; b++) *q++ = p[b];
      k = e;
    } else { *q++ = c; k++; }
  }
}
void launch_gt_render(const uint8_t* u, const uint64_t* src, const uint64_t* off, uint8_t* values, uint64_t N, hipStream_t st) {
  if (!N) return;
  hipLaunchKernelGGL(k_gt_render, dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, st, u, src, off, values, N);
}

void launch_vcf_format_cells(const uint8_t* u, VcfLines L, const uint64_t* rows, uint64_t n, const int32_t* sample_col, int NS,
                             const int16_t* fpos, int S, int gt_field, VcfCellDirect D, uint64_t* sp_off, uint32_t* sp_len,
                             uint8_t* sp_state, uint32_t* err, hipStream_t st) {
  const uint64_t N = n * (uint64_t)NS;
  if (!N || !S) return;
  hipLaunchKernelGGL(k_vcf_format_cells, dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, st, u, L, rows, n, sample_col, NS, fpos, S,
                     gt_field, D, sp_off, sp_len, sp_state, err);
}

// ---- list UDFs (udfs.rs) ------------------------------------------------------------------------------------
__device__ __forceinline__ bool bit_at(const uint64_t* w, uint64_t i) { return w == nullptr || ((w[i >> 6] >> (i & 63)) & 1ull); }
// list_avg: one wave per row.  Int32: exact integer sum (every partial f64 sum of the reference is an exact
// integer below 2^53, so the order of additions cannot change the result).  Float32: lane 0 adds in order.
__global__ __launch_bounds__(256) void k_list_avg(const uint64_t* __restrict__ off, const uint32_t* __restrict__ values,
                                                   const uint64_t* __restrict__ evalid, const uint64_t* __restrict__ lvalid, uint64_t n,
                                                   int is_float, double* __restrict__ out, uint8_t* __restrict__ out_valid) {
  const uint64_t r = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  const uint64_t a = off[r], b = off[r + 1];
  double sum = 0.0;
  unsigned long long cnt = 0;
  if (!bit_at(lvalid, r)) {
    if (lane == 0) { out[r] = 0.0; out_valid[r] = 0; }
    return;
  }
  if (!is_float) {
    long long s = 0;
    for (uint64_t k = a + lane; k < b; k += WAVE)
      if (bit_at(evalid, k)) { s += (int32_t)values[k]; cnt++; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { s += __shfl_down(s, d, 64); cnt += __shfl_down(cnt, d, 64); }
    sum = (double)s;
  } else if (lane == 0) {
    for (uint64_t k = a; k < b; k++)
      if (bit_at(evalid, k)) { sum += (double)__uint_as_float(values[k]); cnt++; }
  }
  if (lane == 0) {
    out[r] = cnt ? sum / (double)cnt : 0.0;
    out_valid[r] = cnt ? 1 : 0;
  }
}
void launch_list_avg(const uint64_t* off, const uint32_t* values, const uint64_t* evalid, const uint64_t* lvalid, uint64_t n,
                     int is_float, double* out, uint8_t* out_valid, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_list_avg, dim3((uint32_t)((n * 64 + 255) / 256)), dim3(256), 0, st, off, values, evalid, lvalid, n, is_float, out,
                     out_valid);
}
// ---- vcf_an / vcf_ac / vcf_af (bio-format-vcf/src/udfs.rs:113-142 parse_gt_alleles, 161-552) ---------------------------------
// One GT string: trimmed; ".", "./." and ".|." are entirely missing; otherwise it is split on '/' and '|', every piece trimmed,
// "." and anything `usize::from_str` rejects (empty, sign other than one leading '+', a non-digit, overflow) is a missing allele.
// `f(idx)` is called for every called allele.  (Rust's trim strips Unicode White_Space; the ASCII members are handled here.)
__device__ __forceinline__ bool gt_is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }
template <typename F>
__device__ __forceinline__ void gt_for_each_allele(const uint8_t* s, uint32_t len, F f) {
  uint32_t a = 0, b = len;
  while (a < b && gt_is_space(s[a])) a++;
  while (b > a && gt_is_space(s[b - 1])) b--;
  const uint32_t n = b - a;
  if (n == 1 && s[a] == '.') return;
  if (n == 3 && s[a] == '.' && s[a + 2] == '.' && (s[a + 1] == '/' || s[a + 1] == '|')) return;
  uint32_t p = a;
  for (;;) {
    uint32_t q = p;
    while (q < b && s[q] != '/' && s[q] != '|') q++;
    uint32_t x = p, y = q;   // the piece [x, y), trimmed
    while (x < y && gt_is_space(s[x])) x++;
    while (y > x && gt_is_space(s[y - 1])) y--;
    if (x < y && s[x] == '+') x++;
    bool ok = x < y;
    unsigned long long v = 0;
    for (uint32_t k = x; k < y && ok; k++) {
      const uint32_t d = (uint32_t)s[k] - '0';
      if (d > 9u) { ok = false; break; }
      if (v > (0xFFFFFFFFFFFFFFFFull - d) / 10ull) { ok = false; break; }   // usize overflow: parse error -> missing
      v = v * 10ull + d;
    }
    if (ok) f(v);
    if (q >= b) break;
    p = q + 1;
  }
}
// per row (one wave): AN = called alleles of its non-NULL GT strings, the largest allele index, both by wave reduction
__global__ __launch_bounds__(256) void k_gt_stats(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ goff,
                                                   const uint64_t* __restrict__ gvalid, const uint64_t* __restrict__ off_g, uint64_t n,
                                                   int32_t* __restrict__ an, unsigned long long* __restrict__ max_allele) {
  const uint64_t r = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  long long cnt = 0;
  unsigned long long mx = 0;
  for (uint64_t k = off_g[r] + lane; k < off_g[r + 1]; k += WAVE) {
    if (!bit_at(gvalid, k)) continue;
    gt_for_each_allele(bytes + goff[k], (uint32_t)(goff[k + 1] - goff[k]), [&](unsigned long long idx) { cnt++; if (idx > mx) mx = idx; });
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    cnt += __shfl_down(cnt, d, 64);
    const unsigned long long o = __shfl_down(mx, d, 64);
    if (o > mx) mx = o;
  }
  if (lane == 0) { an[r] = (int32_t)cnt; max_allele[r] = mx; }
}
// counts[out_off[r] + idx - 1] += 1 for every called allele 1 <= idx <= out_off[r + 1] - out_off[r] of row r (counts zeroed)
__global__ __launch_bounds__(256) void k_gt_ac(const uint8_t* __restrict__ bytes, const uint64_t* __restrict__ goff,
                                                const uint64_t* __restrict__ gvalid, const uint64_t* __restrict__ off_g, uint64_t n,
                                                const uint64_t* __restrict__ out_off, int32_t* __restrict__ counts) {
  const uint64_t r = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  const uint64_t base = out_off[r], vec_len = out_off[r + 1] - base;
  if (!vec_len) return;
  for (uint64_t k = off_g[r] + lane; k < off_g[r + 1]; k += WAVE) {
    if (!bit_at(gvalid, k)) continue;
    gt_for_each_allele(bytes + goff[k], (uint32_t)(goff[k + 1] - goff[k]), [&](unsigned long long idx) {
      if (idx >= 1 && idx <= vec_len) atomicAdd(&counts[base + idx - 1], 1);
    });
  }
}
void launch_gt_stats(const uint8_t* bytes, const uint64_t* goff, const uint64_t* gvalid, const uint64_t* off_g, uint64_t n, int32_t* an,
                     unsigned long long* max_allele, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_gt_stats, dim3((uint32_t)((n * 64 + 255) / 256)), dim3(256), 0, st, bytes, goff, gvalid, off_g, n, an, max_allele);
}
void launch_gt_ac(const uint8_t* bytes, const uint64_t* goff, const uint64_t* gvalid, const uint64_t* off_g, uint64_t n, const uint64_t* out_off,
                  int32_t* counts, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_gt_ac, dim3((uint32_t)((n * 64 + 255) / 256)), dim3(256), 0, st, bytes, goff, gvalid, off_g, n, out_off, counts);
}
// list_gte / list_lte: element-wise compare -> Boolean value bits (element validity and list offsets are the input's)
__global__ __launch_bounds__(256) void k_list_cmp(const uint32_t* __restrict__ values, uint64_t n_elems, int is_float, int op,
                                                   uint32_t thr_bits, uint64_t* __restrict__ out_bits) {
  const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  bool v = false;
  if (c < n_elems) {
    if (is_float) {
      const float x = __uint_as_float(values[c]), t = __uint_as_float(thr_bits);
      v = op == 0 ? x >= t : x <= t;
    } else {
      const int32_t x = (int32_t)values[c], t = (int32_t)thr_bits;
      v = op == 0 ? x >= t : x <= t;
    }
  }
  const unsigned long long m = __ballot(v);
  if ((threadIdx.x & 63) == 0 && (c & ~63ull) < n_elems) out_bits[c >> 6] = m;
}
// counts[0] += popcount(bits & valid), counts[1] += popcount(~valid) over n_elems bits (valid == nullptr: all valid)
__global__ __launch_bounds__(256) void k_count_bits(const uint64_t* __restrict__ bits, const uint64_t* __restrict__ valid, uint64_t n_elems,
                                                     unsigned long long* __restrict__ counts) {
  const uint64_t nw = (n_elems + 63) / 64;
  unsigned long long a = 0, b = 0;
  for (uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x; w < nw; w += (uint64_t)gridDim.x * 256) {
    uint64_t mask = ~0ull;
    if (w == nw - 1 && (n_elems & 63)) mask = (1ull << (n_elems & 63)) - 1;
    const uint64_t v = valid ? valid[w] : ~0ull;
    a += (unsigned long long)__popcll(bits[w] & v & mask);
    b += (unsigned long long)__popcll(~v & mask);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { a += __shfl_down(a, d, 64); b += __shfl_down(b, d, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&counts[0], a); atomicAdd(&counts[1], b); }
}
void launch_count_bits(const uint64_t* bits, const uint64_t* valid, uint64_t n_elems, unsigned long long* counts, hipStream_t st) {
  if (!n_elems) return;
  const uint64_t nw = (n_elems + 63) / 64;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((nw + 255) / 256, 2048);
  hipLaunchKernelGGL(k_count_bits, dim3(grid), dim3(256), 0, st, bits, valid, n_elems, counts);
}
// list_and (udfs.rs:799-843): one wave per row, SQL three-valued AND over min(len_l, len_r) elements.  The output
// bit range of a 64-element step is unaligned, so a wave ORs its ballot into the two words it straddles
// (out_val / out_valid must be zeroed first).
__device__ __forceinline__ void or_bits(uint64_t* words, uint64_t bit0, unsigned long long m, int lane) {
  if (lane == 0 && m) {
    const uint32_t sh = (uint32_t)(bit0 & 63);
    atomicOr((unsigned long long*)&words[bit0 >> 6], m << sh);
    if (sh && (m >> (64 - sh))) atomicOr((unsigned long long*)&words[(bit0 >> 6) + 1], m >> (64 - sh));
  }
}
__global__ __launch_bounds__(256) void k_list_and(const uint64_t* __restrict__ off_l, const uint64_t* __restrict__ off_r,
                                                   const uint64_t* __restrict__ off_o, const uint64_t* __restrict__ lval,
                                                   const uint64_t* __restrict__ lvalid, const uint64_t* __restrict__ rval,
                                                   const uint64_t* __restrict__ rvalid, uint64_t n, uint64_t* __restrict__ out_val,
                                                   uint64_t* __restrict__ out_valid) {
  const uint64_t r = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  const uint64_t a = off_l[r], b = off_r[r], o = off_o[r];
  const uint64_t len = off_o[r + 1] - o;
  for (uint64_t j0 = 0; j0 < len; j0 += WAVE) {
    const uint64_t j = j0 + lane;
    bool v = false, ok = false;
    if (j < len) {
      const bool ln = !bit_at(lvalid, a + j), rn = !bit_at(rvalid, b + j);
      const bool lv = ((lval[(a + j) >> 6] >> ((a + j) & 63)) & 1ull) != 0, rv = ((rval[(b + j) >> 6] >> ((b + j) & 63)) & 1ull) != 0;
      if (ln && rn) ok = false;
      else if (ln) ok = !rv;          // NULL AND false = false, NULL AND true = NULL
      else if (rn) ok = !lv;
      else { ok = true; v = lv && rv; }
    }
    or_bits(out_val, o + j0, __ballot(v), lane);
    or_bits(out_valid, o + j0, __ballot(ok), lane);
  }
}
void launch_list_and(const uint64_t* off_l, const uint64_t* off_r, const uint64_t* off_o, const uint64_t* lval, const uint64_t* lvalid,
                     const uint64_t* rval, const uint64_t* rvalid, uint64_t n, uint64_t* out_val, uint64_t* out_valid, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_list_and, dim3((uint32_t)((n * 64 + 255) / 256)), dim3(256), 0, st, off_l, off_r, off_o, lval, lvalid, rval,
                     rvalid, n, out_val, out_valid);
}
// vcf_set_gts (udfs.rs:896-949): per GT element the output length / source / validity.  src indexes the buffer
// [GT value bytes | replacement]: a replaced element points at the replacement (rep_off).
__global__ __launch_bounds__(256) void k_set_gts_plan(const uint64_t* __restrict__ off_g, const uint64_t* __restrict__ goff,
                                                       const uint64_t* __restrict__ gvalid, const uint64_t* __restrict__ off_m,
                                                       const uint64_t* __restrict__ mlvalid, const uint64_t* __restrict__ mval,
                                                       const uint64_t* __restrict__ mvalid, uint64_t n, uint64_t rep_off, uint32_t rep_len,
                                                       uint64_t* __restrict__ src, uint32_t* __restrict__ len, uint8_t* __restrict__ ovalid) {
  const uint64_t r = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  const uint64_t a = off_g[r], e = off_g[r + 1];
  const bool mnull = !bit_at(mlvalid, r);
  const uint64_t mb = off_m[r], mlen = off_m[r + 1] - mb;
  for (uint64_t j = lane; a + j < e; j += WAVE) {
    const uint64_t g = a + j;
    bool keep = true;
    if (!mnull && j < mlen && bit_at(mvalid, mb + j)) keep = ((mval[(mb + j) >> 6] >> ((mb + j) & 63)) & 1ull) != 0;
    if (keep) {
      const bool gv = bit_at(gvalid, g);
      src[g] = goff[g];
      len[g] = gv ? (uint32_t)(goff[g + 1] - goff[g]) : 0u;
      ovalid[g] = gv ? 1 : 0;
    } else {
      src[g] = rep_off;
      len[g] = rep_len;
      ovalid[g] = 1;
    }
  }
}
void launch_set_gts_plan(const uint64_t* off_g, const uint64_t* goff, const uint64_t* gvalid, const uint64_t* off_m,
                         const uint64_t* mlvalid, const uint64_t* mval, const uint64_t* mvalid, uint64_t n, uint64_t rep_off,
                         uint32_t rep_len, uint64_t* src, uint32_t* len, uint8_t* ovalid, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_set_gts_plan, dim3((uint32_t)((n * 64 + 255) / 256)), dim3(256), 0, st, off_g, goff, gvalid, off_m, mlvalid, mval,
                     mvalid, n, rep_off, rep_len, src, len, ovalid);
}
void launch_list_cmp(const uint32_t* values, uint64_t n_elems, int is_float, int op, uint32_t thr_bits, uint64_t* out_bits, hipStream_t st) {
  if (!n_elems) return;
  hipLaunchKernelGGL(k_list_cmp, dim3((uint32_t)((n_elems + 255) / 256)), dim3(256), 0, st, values, n_elems, is_float, op, thr_bits, out_bits);
}

}  // namespace bioscan
