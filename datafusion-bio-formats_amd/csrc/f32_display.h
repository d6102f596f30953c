// Rust `f32::to_string()` on device (an 'f' aux value landing in a Utf8 tag column, sam_tag_io.rs:703-760 via
// `value.to_string()`): the shortest decimal digit string that round-trips to the same f32, closest to the exact value,
// printed positionally without an exponent ("0.0000001", "100000000000000000000"), "NaN", "inf", "-inf", "-0".
// Free-format digit generation (Steele & White / Burger & Dybvig) on 224-bit integers: exact, no tables.  This is a
// corner path (a float in a string column), so it favours being obviously exact over speed; it is kept out of line so
// the callers' register budgets are not affected.
#pragma once
#include <stdint.h>
#include <math.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define F32D_FN __host__ __device__ inline
#define F32D_ENTRY __host__ __device__ __noinline__
#else  // plain g++: the CPU unit test (tests/test_cpu_host_logic.py) compiles this header to check it against numpy
#define F32D_FN inline
#define F32D_ENTRY inline
#endif

namespace f32disp {
constexpr int LIMBS = 7;
struct Big { uint32_t w[LIMBS]; };

F32D_FN void big_set(Big& a, uint32_t v) { a.w[0] = v; for (int i = 1; i < LIMBS; i++) a.w[i] = 0; }
F32D_FN void big_mul_small(Big& a, uint32_t m) {
  uint64_t c = 0;
  for (int i = 0; i < LIMBS; i++) { c += (uint64_t)a.w[i] * m; a.w[i] = (uint32_t)c; c >>= 32; }
}
F32D_FN void big_shl(Big& a, uint32_t s) {
  const uint32_t q = s >> 5, r = s & 31;
  for (int i = LIMBS - 1; i >= 0; i--) {
    uint32_t hi = (i >= (int)q) ? a.w[i - q] : 0u;
    uint32_t lo = (r && i >= (int)q + 1) ? a.w[i - q - 1] : 0u;
    a.w[i] = r ? ((hi << r) | (lo >> (32 - r))) : hi;
  }
}
F32D_FN int big_cmp(const Big& a, const Big& b) {
  for (int i = LIMBS - 1; i >= 0; i--) if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
  return 0;
}
F32D_FN void big_add(Big& a, const Big& b) {
  uint64_t c = 0;
  for (int i = 0; i < LIMBS; i++) { c += (uint64_t)a.w[i] + b.w[i]; a.w[i] = (uint32_t)c; c >>= 32; }
}
F32D_FN void big_sub(Big& a, const Big& b) {  // a >= b
  int64_t c = 0;
  for (int i = 0; i < LIMBS; i++) { c += (int64_t)a.w[i] - b.w[i]; a.w[i] = (uint32_t)c; c >>= 32; }
}
F32D_FN void big_mul_pow10(Big& a, uint32_t k) {
  while (k >= 9) { big_mul_small(a, 1000000000u); k -= 9; }
  uint32_t m = 1;
  while (k--) m *= 10;
  if (m != 1) big_mul_small(a, m);
}

// Writes the display form of the f32 with bit pattern `bits` to out (needs up to 50 bytes), returns its length.
F32D_ENTRY uint32_t f32_display(uint32_t bits, uint8_t* out) {
  const bool neg = bits >> 31;
  const uint32_t be = (bits >> 23) & 0xFF, frac = bits & 0x7FFFFF;
  uint32_t n = 0;
  if (be == 0xFF) {
    if (frac) { out[0] = 'N'; out[1] = 'a'; out[2] = 'N'; return 3; }
    if (neg) out[n++] = '-';
    out[n++] = 'i'; out[n++] = 'n'; out[n++] = 'f';
    return n;
  }
  if (neg) out[n++] = '-';
  if (be == 0 && frac == 0) { out[n++] = '0'; return n; }
  const uint32_t m = be ? (frac | 0x800000u) : frac;
  const int e = be ? (int)be - 150 : -149;
  const bool even = !(m & 1);
  const bool boundary = (frac == 0 && be > 1);  // the gap below is half the gap above
  // v = r / s, upper half-gap = mp / s, lower half-gap = mm / s
  Big r, s, mp, mm;
  big_set(r, m); big_set(s, 1); big_set(mp, 1); big_set(mm, 1);
  if (e >= 0) {
    big_shl(r, (uint32_t)e + (boundary ? 2 : 1));
    big_shl(s, boundary ? 2 : 1);
    big_shl(mp, (uint32_t)e + (boundary ? 1 : 0));
    big_shl(mm, (uint32_t)e);
  } else {
    big_shl(r, boundary ? 2 : 1);
    big_shl(s, (uint32_t)(-e) + (boundary ? 2 : 1));
    if (boundary) big_shl(mp, 1);
  }
  const int p = 32 - __builtin_clz(m);  // bit length of m
  int k = (int)ceil((double)(e + p - 1) * 0.30102999566398120 - 1e-10);
  if (k >= 0) big_mul_pow10(s, (uint32_t)k);
  else { big_mul_pow10(r, (uint32_t)(-k)); big_mul_pow10(mp, (uint32_t)(-k)); big_mul_pow10(mm, (uint32_t)(-k)); }
  // fix-up: k is the smallest integer with (r + mp) / s below (or at, when the boundary itself does not round to v) 1
  for (int guard = 0; guard < 3; guard++) {
    Big t = r; big_add(t, mp);
    const int c = big_cmp(t, s);
    if (even ? c >= 0 : c > 0) { big_mul_small(s, 10); k++; } else break;
  }
  uint8_t dig[12];
  int nd = 0;
  for (;;) {
    big_mul_small(r, 10); big_mul_small(mp, 10); big_mul_small(mm, 10);
    uint32_t d = 0;
    while (big_cmp(r, s) >= 0) { big_sub(r, s); d++; }
    const int cl = big_cmp(r, mm);
    const bool low = even ? cl <= 0 : cl < 0;
    Big t = r; big_add(t, mp);
    const int ch = big_cmp(t, s);
    const bool high = even ? ch >= 0 : ch > 0;
    if (!low && !high && nd < 10) { dig[nd++] = (uint8_t)d; continue; }
    if (low && high) {  // both neighbours round-trip: take the closer one (2r vs s; a tie cannot occur for binary32)
      Big t2 = r; big_shl(t2, 1);
      if (big_cmp(t2, s) >= 0) d++;
    } else if (high) d++;
    dig[nd++] = (uint8_t)d;
    break;
  }
  // a final digit of 10 carries
  for (int i = nd - 1; i >= 0 && dig[i] == 10; i--) {
    dig[i] = 0;
    if (i) dig[i - 1]++;
    else { for (int j = nd; j > 0; j--) dig[j] = dig[j - 1]; dig[0] = 1; nd++; k++; }
  }
  while (nd > 1 && dig[nd - 1] == 0) nd--;
  // value = 0.d1 d2 ... x 10^k
  if (k <= 0) {
    out[n++] = '0'; out[n++] = '.';
    for (int i = 0; i < -k; i++) out[n++] = '0';
    for (int i = 0; i < nd; i++) out[n++] = (uint8_t)('0' + dig[i]);
  } else if (k < nd) {
    for (int i = 0; i < k; i++) out[n++] = (uint8_t)('0' + dig[i]);
    out[n++] = '.';
    for (int i = k; i < nd; i++) out[n++] = (uint8_t)('0' + dig[i]);
  } else {
    for (int i = 0; i < nd; i++) out[n++] = (uint8_t)('0' + dig[i]);
    for (int i = nd; i < k; i++) out[n++] = '0';
  }
  return n;
}
}  // namespace f32disp
