// bam_write.hip -- the write side of the BAM path on gfx950: Arrow columns -> BAM records -> BGZF members.
//
// Replaces, for local BAM output, the reference's batch_to_alignment_records
// (bio-format-core/src/sam_record_serializer.rs:15-258) + noodles-bam's record encoder + noodles-bgzf's Writer
// (bio-format-bam/src/writer.rs, serializer.rs, write_exec.rs).  Three stages, all integer / byte work (no MFMA):
//   serialize  one row per lane: record sizes (pass 1), exclusive scan, record bytes (pass 2) -- header fields, read name,
//              CIGAR parsed from its string (or copied when binary), 4-bit packed bases, qualities minus 33;
//   crc32      one member per lane (k_bgzf_crc32 in store mode);
//   deflate    one BGZF member (<= 65280 payload bytes) per wavefront, two passes: (1) parse -- 64 positions per step, a
//              two-way 4096-entry hash table of 3-byte prefixes in LDS proposes two candidates per lane (+ the distance-1
//              candidate for runs), lanes measure their matches, a match yields to a longer one at the next position, a
//              scalar walk picks the parse of the 64 positions; the tokens
//              go to a scratch list and the symbols are counted in LDS; the block's OWN length-limited Huffman codes are
//              built (literal/length, distance, and the code-length code of the header); (2) the tokens are coded -- bit
//              offsets from a wave prefix sum, bits OR-ed into an LDS staging window that is flushed as whole dwords.  The
//              cheaper of the dynamic block (header included) and the FIXED code (RFC 1951 3.2.6) is written; a member whose
//              coded form would be larger than the payload is written as a stored block, so a member never exceeds 64 KiB.
// Every member is valid BGZF (gzip header with the BC subfield, CRC32, ISIZE): zlib, libdeflate and K1 read it back.
#include "kernels.h"

namespace bioscan {

#define WAVE 64

struct __attribute__((packed, aligned(1))) bw_u32 { uint32_t v; };
struct __attribute__((packed, aligned(1))) bw_u64 { uint64_t v; };
struct __attribute__((packed, aligned(1))) bw_u16 { uint16_t v; };
__device__ __forceinline__ void bw_st32(uint8_t* p, uint32_t v) { ((bw_u32*)p)->v = v; }
__device__ __forceinline__ void bw_st16(uint8_t* p, uint32_t v) { ((bw_u16*)p)->v = (uint16_t)v; }

__device__ __forceinline__ bool bit_valid(const uint8_t* bits, int64_t i) { return !bits || ((bits[i >> 3] >> (i & 7)) & 1); }

// =================================================================================================================
// serializer
// =================================================================================================================
__device__ __forceinline__ int cigar_op_code(uint8_t c) {
  switch (c) {
    case 'M': return 0; case 'I': return 1; case 'D': return 2; case 'N': return 3; case 'S': return 4;
    case 'H': return 5; case 'P': return 6; case '=': return 7; case 'X': return 8;
    default: return -1;
  }
}
// number of ops of a CIGAR string (-1: malformed); "*" and "" are the empty CIGAR (sam_record_serializer.rs:225-237)
__device__ int cigar_count_ops(const uint8_t* s, int32_t l) {
  if (l == 0 || (l == 1 && s[0] == '*')) return 0;
  int n = 0;
  bool digits = false;
  for (int32_t k = 0; k < l; k++) {
    const uint8_t c = s[k];
    if (c >= '0' && c <= '9') { digits = true; continue; }
    if (!digits || cigar_op_code(c) < 0) return -1;
    digits = false;
    n++;
  }
  return digits ? -1 : n;
}

// samtools / SAM spec 5.3 reg2bin on a 0-based half-open interval
__device__ __forceinline__ uint32_t reg2bin(int64_t beg, int64_t end) {
  --end;
  if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}

struct SerRow {
  uint32_t lrn, ncig, lseq, lqual;
  bool star_name;
  const uint8_t *name, *cigar, *seq, *qual;
  int32_t lname, lcigar;
};
__device__ __forceinline__ SerRow ser_row(const SerCols& c, uint64_t i, uint32_t* err) {
  SerRow r;
  const int64_t j = (int64_t)i + c.offset;
  const int32_t n0 = c.name_off[j], n1 = c.name_off[j + 1];
  r.name = c.name + n0; r.lname = n1 - n0;
  // a NULL name and "*" are the missing name (sam_record_serializer.rs:131-135): "*\0" on disk
  r.star_name = !bit_valid(c.name_valid, j) || (r.lname == 1 && r.name[0] == '*') || r.lname == 0;
  r.lrn = r.star_name ? 2u : (uint32_t)r.lname + 1u;
  if (r.lrn > 255u) atomicExch(err, 4u);
  const int32_t c0 = c.cigar_off[j], c1 = c.cigar_off[j + 1];
  r.cigar = c.cigar + c0; r.lcigar = c1 - c0;
  if (c.cigar_binary) {
    if (r.lcigar & 3) atomicExch(err, 2u);
    r.ncig = (uint32_t)r.lcigar >> 2;
  } else {
    const int n = cigar_count_ops(r.cigar, r.lcigar);
    if (n < 0) atomicExch(err, 2u);
    r.ncig = n < 0 ? 0u : (uint32_t)n;
  }
  if (r.ncig > 65535u) atomicExch(err, 5u);
  const int32_t s0 = c.seq_off[j], s1 = c.seq_off[j + 1];
  r.seq = c.seq + s0;
  r.lseq = (uint32_t)(s1 - s0);
  if (r.lseq == 1 && r.seq[0] == '*') r.lseq = 0;
  const int32_t q0 = c.qual_off[j], q1 = c.qual_off[j + 1];
  r.qual = c.qual + q0;
  r.lqual = (uint32_t)(q1 - q0);
  if (r.lqual == 1 && r.qual[0] == '*') r.lqual = 0;
  // noodles' encoder: qualities are either absent (0xFF fill) or exactly one per base
  if (r.lqual != 0 && r.lqual != r.lseq) atomicExch(err, 3u);
  if (c.flags[j] > 65535u) atomicExch(err, 1u);  // "does not fit into 16-bit SAM flags" (serializer.rs:117-122)
  return r;
}

// ---- tag columns -> aux fields (sam_tag_io.rs:109-147, 206-656; noodles-bam's data encoder) -------------------------
__device__ __forceinline__ uint32_t sam_int_width(uint8_t t) { return (t == 'c' || t == 'C') ? 1u : (t == 's' || t == 'S') ? 2u : 4u; }
// the element at index j of a fixed-width buffer as (signed, value); false for a float / other kind
__device__ __forceinline__ bool ser_int_at(uint8_t kind, const uint8_t* v, int64_t j, bool* neg, uint64_t* mag_or_val, int64_t* sv) {
  switch (kind) {
    case SK_I8: *sv = ((const int8_t*)v)[j]; break;
    case SK_I16: *sv = ((const int16_t*)v)[j]; break;
    case SK_I32: *sv = ((const int32_t*)v)[j]; break;
    case SK_I64: *sv = ((const int64_t*)v)[j]; break;
    case SK_U8: *mag_or_val = ((const uint8_t*)v)[j]; *neg = false; *sv = 0; return true;
    case SK_U16: *mag_or_val = ((const uint16_t*)v)[j]; *neg = false; *sv = 0; return true;
    case SK_U32: *mag_or_val = ((const uint32_t*)v)[j]; *neg = false; *sv = 0; return true;
    case SK_U64: *mag_or_val = ((const uint64_t*)v)[j]; *neg = false; *sv = 0; return true;
    default: return false;
  }
  *neg = *sv < 0;
  *mag_or_val = (uint64_t)*sv;
  return true;
}
// T::try_from(value) for the SAM integer types; the accepted value's low bytes are the encoding
__device__ __forceinline__ bool sam_int_fits(uint8_t t, bool neg, uint64_t uv, int64_t sv) {
  switch (t) {
    case 'c': return neg ? sv >= -128 : uv <= 127u;
    case 's': return neg ? sv >= -32768 : uv <= 32767u;
    case 'i': return neg ? sv >= -2147483648ll : uv <= 2147483647u;
    case 'C': return !neg && uv <= 255u;
    case 'S': return !neg && uv <= 65535u;
    case 'I': return !neg && uv <= 4294967295ull;
    default: return false;
  }
}
__device__ __forceinline__ bool f64_fits_f32(double v) { return isfinite(v) && v >= -3.4028234663852886e38 && v <= 3.4028234663852886e38; }
__device__ __forceinline__ uint8_t ascii_upper(uint8_t b) { return (b >= 'a' && b <= 'z') ? (uint8_t)(b - 32) : b; }

// Bytes of one tag column's aux field for the row (0: NULL, nothing is written); o == nullptr only measures and reports.
__device__ uint32_t ser_tag_field(const SerTagCol& t, uint64_t row, uint32_t ci, uint8_t* o, unsigned long long* tag_err) {
  const int64_t j = (int64_t)row + t.offset;
  if (!bit_valid(t.valid, j)) return 0;
  auto fail = [&](uint32_t code) {
    if (tag_err) atomicMin(tag_err, ((unsigned long long)row << 16) | ((unsigned long long)ci << 8) | code);
    return 0u;
  };
  auto head = [&](uint8_t type) { if (o) { o[0] = t.tag[0]; o[1] = t.tag[1]; o[2] = type; } };
  bool neg = false;
  uint64_t uv = 0;
  int64_t sv = 0;
  switch (t.sam_type) {
    case 'i': case 'c': case 's': case 'C': case 'S': case 'I': {
      if (!ser_int_at(t.kind, t.values, j, &neg, &uv, &sv)) return 0;  // (the host has rejected the column's type)
      if (!sam_int_fits(t.sam_type, neg, uv, sv)) return fail(20);
      const uint32_t w = sam_int_width(t.sam_type);
      head(t.sam_type);
      if (o) for (uint32_t k = 0; k < w; k++) o[3 + k] = (uint8_t)(uv >> (8 * k));
      return 3 + w;
    }
    case 'f': {
      float f;
      if (t.kind == SK_F32) f = ((const float*)t.values)[j];
      else { const double d = ((const double*)t.values)[j]; if (!f64_fits_f32(d)) return fail(21); f = (float)d; }
      head('f');
      if (o) bw_st32(o + 3, __float_as_uint(f));
      return 7;
    }
    case 'A': {
      uint8_t b;
      if (t.kind == SK_UTF8) {
        const int32_t a = t.off[j], e = t.off[j + 1];
        if (e - a != 1 || t.values[a] >= 128) return fail(23);
        b = t.values[a];
      } else {
        if (!ser_int_at(t.kind, t.values, j, &neg, &uv, &sv)) return 0;
        if (neg || uv > 255u) return fail(24);
        b = (uint8_t)uv;
      }
      head('A');
      if (o) o[3] = b;
      return 4;
    }
    case 'H': {
      const int32_t a = t.off[j], e = t.off[j + 1];
      if ((e - a) & 1) return fail(22);
      for (int32_t k = a; k < e; k++) {
        const uint8_t b = ascii_upper(t.values[k]);
        if (!((b >= '0' && b <= '9') || (b >= 'A' && b <= 'F'))) return fail(22);
        if (o) o[3 + (k - a)] = b;
      }
      head('H');
      if (o) o[3 + (e - a)] = 0;
      return 4 + (uint32_t)(e - a);
    }
    case 'B': {
      const int32_t a = t.off[j], e = t.off[j + 1];
      const uint32_t cnt = (uint32_t)(e - a);
      const uint32_t w = t.subtype == 'f' ? 4u : sam_int_width(t.subtype);
      head('B');
      if (o) { o[3] = t.subtype; bw_st32(o + 4, cnt); }
      for (int32_t k = a; k < e; k++) {
        const int64_t q = (int64_t)k + t.eoffset;
        if (!bit_valid(t.evalid, q)) return fail(25);
        uint32_t bits;
        if (t.subtype == 'f') {
          if (t.ekind == SK_F32) bits = __float_as_uint(((const float*)t.values)[q]);
          else { const double d = ((const double*)t.values)[q]; if (!f64_fits_f32(d)) return fail(26); bits = __float_as_uint((float)d); }
        } else {
          if (!ser_int_at(t.ekind, t.values, q, &neg, &uv, &sv)) return 0;
          if (!sam_int_fits(t.subtype, neg, uv, sv)) return fail(26);
          bits = (uint32_t)uv;
        }
        if (o) for (uint32_t b = 0; b < w; b++) o[8 + (uint32_t)(k - a) * w + b] = (uint8_t)(bits >> (8 * b));
      }
      return 8 + cnt * w;
    }
    default: {  // 'Z', and any other type character on a string column (sam_tag_io.rs:227-233)
      const int32_t a = t.off[j], e = t.off[j + 1];
      head('Z');
      if (o) { for (int32_t k = a; k < e; k++) o[3 + (k - a)] = t.values[k]; o[3 + (e - a)] = 0; }
      return 4 + (uint32_t)(e - a);
    }
  }
}

__global__ __launch_bounds__(256) void k_ser_sizes(SerCols c, SerTags tg, uint64_t n, uint32_t* __restrict__ rec_bytes, uint32_t* err,
                                                   unsigned long long* tag_err) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const SerRow r = ser_row(c, i, err);
  uint32_t aux = 0;
  for (int32_t k = 0; k < tg.n; k++) aux += ser_tag_field(tg.cols[k], i, (uint32_t)k, nullptr, tag_err);
  rec_bytes[i] = 4u + 32u + r.lrn + 4u * r.ncig + ((r.lseq + 1u) >> 1) + r.lseq + aux;
}

__global__ __launch_bounds__(256) void k_ser_write(SerCols c, SerTags tg, uint64_t n, const uint64_t* __restrict__ rec_off,
                                                   uint8_t* __restrict__ out, uint32_t* err) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t dummy = 0;
  (void)dummy;
  const SerRow r = ser_row(c, i, err);
  const int64_t j = (int64_t)i + c.offset;
  uint8_t* o = out + rec_off[i];
  const uint32_t block_size = (uint32_t)(rec_off[i + 1] - rec_off[i]) - 4u;  // aux fields included
  const int32_t refid = c.refid[i], nref = c.mate_refid[i];
  int32_t pos = -1, npos = -1;
  if (bit_valid(c.start_valid, j)) {
    const uint32_t v = c.start[j];
    const int64_t p1 = c.zero_based ? (int64_t)v + 1 : (int64_t)v;   // 1-based; 0 is not a position (None)
    pos = p1 >= 1 ? (int32_t)(p1 - 1) : -1;
  }
  if (bit_valid(c.mate_start_valid, j)) {
    const uint32_t v = c.mate_start[j];
    const int64_t p1 = c.zero_based ? (int64_t)v + 1 : (int64_t)v;
    npos = p1 >= 1 ? (int32_t)(p1 - 1) : -1;
  }
  // CIGAR ops (and the reference span for the bin)
  uint8_t* cg = o + 36 + r.lrn;
  uint64_t span = 0;
  if (c.cigar_binary) {
    for (uint32_t k = 0; k < r.ncig; k++) {
      const uint32_t v = ((const bw_u32*)(r.cigar + 4 * k))->v;
      // decode_binary_cigar_to_ops (alignment_utils.rs:985-1017): op codes 0..8, a zero length is invalid
      if ((v & 15u) > 8u || (v >> 4) == 0u) atomicExch(err, 2u);
      if ((0x18Du >> (v & 15u)) & 1u) span += v >> 4;
      bw_st32(cg + 4 * k, v);
    }
  } else {
    uint32_t k = 0, num = 0;
    for (int32_t q = 0; q < r.lcigar && k < r.ncig; q++) {
      const uint8_t ch = r.cigar[q];
      if (ch >= '0' && ch <= '9') { num = num * 10u + (uint32_t)(ch - '0'); continue; }
      const uint32_t op = (uint32_t)cigar_op_code(ch);
      if (num > 0x0FFFFFFFu) atomicExch(err, 2u);
      if ((0x18Du >> op) & 1u) span += num;
      bw_st32(cg + 4 * k, (num << 4) | op);
      k++;
      num = 0;
    }
  }
  const uint32_t bin = pos >= 0 ? reg2bin(pos, (int64_t)pos + (int64_t)(span ? span : 1)) : 4680u;
  bw_st32(o, block_size);
  bw_st32(o + 4, (uint32_t)refid);
  bw_st32(o + 8, (uint32_t)pos);
  o[12] = (uint8_t)r.lrn;
  o[13] = (uint8_t)c.mapq[j];          // MappingQuality::new(v as u8): 255 = missing, written as 255
  bw_st16(o + 14, bin);
  bw_st16(o + 16, r.ncig);
  bw_st16(o + 18, c.flags[j]);
  bw_st32(o + 20, r.lseq);
  bw_st32(o + 24, (uint32_t)nref);
  bw_st32(o + 28, (uint32_t)npos);
  bw_st32(o + 32, (uint32_t)c.tlen[j]);
  uint8_t* nm = o + 36;
  if (r.star_name) { nm[0] = '*'; nm[1] = 0; }
  else { for (int32_t k = 0; k < r.lname; k++) nm[k] = r.name[k]; nm[r.lname] = 0; }
  // bases: "=ACMGRSVTWYHKDBN", case-insensitive, anything else is N (noodles' sequence encoder)
  uint8_t* sq = cg + 4 * r.ncig;
  auto code = [](uint8_t b) -> uint32_t {
    switch (b & 0xDF) {  // upper-case
      case 'A': return 1; case 'C': return 2; case 'M': return 3; case 'G': return 4; case 'R': return 5; case 'S': return 6;
      case 'V': return 7; case 'T': return 8; case 'W': return 9; case 'Y': return 10; case 'H': return 11; case 'K': return 12;
      case 'D': return 13; case 'B': return 14; case 'N': return 15;
      default: return b == '=' ? 0u : 15u;
    }
  };
  for (uint32_t k = 0; k < r.lseq; k += 2) {
    const uint32_t hi = code(r.seq[k]);
    const uint32_t lo = k + 1 < r.lseq ? code(r.seq[k + 1]) : 0u;
    sq[k >> 1] = (uint8_t)((hi << 4) | lo);
  }
  uint8_t* ql = sq + ((r.lseq + 1u) >> 1);
  if (r.lqual == 0) { for (uint32_t k = 0; k < r.lseq; k++) ql[k] = 0xFF; }
  else { for (uint32_t k = 0; k < r.lseq; k++) { const uint8_t b = r.qual[k]; ql[k] = b >= 33 ? (uint8_t)(b - 33) : 0; } }  // saturating_sub(33)
  uint8_t* ax = ql + r.lseq;
  for (int32_t k = 0; k < tg.n; k++) ax += ser_tag_field(tg.cols[k], i, (uint32_t)k, ax, nullptr);
}

void launch_ser_sizes(SerCols c, SerTags t, uint64_t n, uint32_t* rec_bytes, uint32_t* err, unsigned long long* tag_err, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_ser_sizes, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, c, t, n, rec_bytes, err, tag_err);
}
void launch_ser_write(SerCols c, SerTags t, uint64_t n, const uint64_t* rec_off, uint8_t* out, uint32_t* err, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_ser_write, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, c, t, n, rec_off, out, err);
}

// =================================================================================================================
// BGZF deflate: one member per wavefront
// =================================================================================================================
#ifndef DF_HASH_BITS_N
#define DF_HASH_BITS_N 11
#endif
constexpr int DF_HASH_BITS = DF_HASH_BITS_N;
#ifndef DF_WAYS
#define DF_WAYS 6            // candidates per hash (the most recent positions): 6 x 2048 entries = 24 KB of LDS
#endif
// A short match far away costs more bits than the literals it replaces (zlib's TOO_FAR): length 3 only within DF_FAR3 bytes,
// length 4 within DF_FAR4.  On config-2 payload (CPU model of this parse, tools/experiments/w2_parse_model.py): 2 ways 0.451 ->
// 0.436 with the two rules alone, 4 ways 0.430, 8 ways 0.427; 64 ways WITHOUT them 0.437 -- the rules are worth more than
// any number of candidates.  Measured on the GPU (tools/w2_variants.sh; 266 MB of config-2 payload; r03: 2 ways x 4096 without the
// rules 0.4440 at 16.4 GB/s): 2 ways 0.4278 / 15.1 GB/s, 3 ways 0.4242 / 13.7, 4 x 2048 0.4239 / 12.0, 4 x 4096 0.4222 / 8.9 (LDS:
// one wave per SIMD), **6 x 2048 0.4212 / 10.1 (the default)**, 8 x 2048 0.4199 / 6.2; zlib -6: 0.3949.
#ifndef DF_FAR3
#define DF_FAR3 128u
#endif
#ifndef DF_FAR4
#define DF_FAR4 4096u
#endif
constexpr uint32_t DF_NONE = 0xFFFFu;
constexpr int DF_WORDS = 160;  // staging window: a carried partial dword + 64 tokens of <= 48 bits (97 dwords); the block header (<= 75 dwords)
constexpr int DF_NLIT = 286, DF_NDIST = 30, DF_NPRE = 19;
constexpr uint32_t DF_TOK_MATCH = 0x80000000u;  // token: literal = byte (256 = END-OF-BLOCK); match = flag | (dist - 1) << 8 | (len - 3)

__device__ __forceinline__ uint32_t df_excl_scan(uint32_t v, int lane, uint32_t* total) {
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, WAVE);
    if (lane >= d) inc += o;
  }
  *total = __builtin_amdgcn_readlane(inc, 63);
  return inc - v;
}
__device__ __forceinline__ uint32_t df_wave_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, WAVE);
  return v;
}

// common prefix length of a[0..) and b[0..), at most `cap` (reads up to 7 bytes past cap: buffers are padded)
__device__ __forceinline__ uint32_t df_match_len(const uint8_t* a, const uint8_t* b, uint32_t cap) {
  uint32_t k = 0;
  while (k < cap) {
    const uint64_t x = ((const bw_u64*)(a + k))->v ^ ((const bw_u64*)(b + k))->v;
    if (x) { k += (uint32_t)__builtin_ctzll(x) >> 3; break; }
    k += 8;
  }
  return k < cap ? k : cap;
}

// length / distance -> symbol, number of extra bits, extra bits (RFC 1951 3.2.5)
__device__ __forceinline__ void df_len_sym(uint32_t len, uint32_t* sym, uint32_t* eb, uint32_t* ex) {
  const uint32_t lc = len - 3;
  if (lc == 255) { *sym = 285; *eb = 0; *ex = 0; }
  else if (lc < 8) { *sym = 257 + lc; *eb = 0; *ex = 0; }
  else { const uint32_t b = (31u - (uint32_t)__builtin_clz(lc)) - 2u; *eb = b; *sym = 261 + 4 * b + ((lc >> b) & 3u); *ex = lc & ((1u << b) - 1u); }
}
__device__ __forceinline__ void df_dist_sym(uint32_t dist, uint32_t* sym, uint32_t* eb, uint32_t* ex) {
  const uint32_t dc = dist - 1;
  if (dc < 4) { *sym = dc; *eb = 0; *ex = 0; }
  else { const uint32_t b = (31u - (uint32_t)__builtin_clz(dc)) - 1u; *eb = b; *sym = 2 * b + 2 + ((dc >> b) & 1u); *ex = dc & ((1u << b) - 1u); }
}

struct DfLds {
  uint16_t table[DF_WAYS][1 << DF_HASH_BITS];   // the DF_WAYS most recent positions of every hash
  uint32_t W[DF_WORDS];
  uint32_t hl[DF_NLIT + 2], hd[DF_NDIST + 2], hp[DF_NPRE + 1];  // symbol counts
  uint32_t cl[DF_NLIT + 2], cd[DF_NDIST + 2], cp[DF_NPRE + 1];  // bit-reversed code << 8 | length
  uint16_t sorted[DF_NLIT + 2];
  uint32_t weight[2 * DF_NLIT];
  uint16_t parent[2 * DF_NLIT];
  uint8_t len[DF_NLIT + 2];
  uint16_t hsym[DF_NLIT + DF_NDIST + 4];  // block header: code-length symbol | extra bits << 5 | number of extra bits << 12
  uint32_t nh, ok;
};
__device__ __forceinline__ void df_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Length-limited canonical Huffman code of freq[0 .. n): code[s] = bit-reversed code << 8 | length (0 for unused symbols).
// The symbols are ranked in parallel; the tree (two-queue merge over the sorted leaves), the depths, the length limit
// (over-long codes are clamped, the Kraft sum is repaired by lengthening the longest shorter codes, what is left over is given
// back to the most frequent symbols of each length) and the canonical numbering run on lane 0: a few microseconds per
// member, beside ~1000 parse steps.  A one-symbol alphabet gets a second, unused symbol so that the code is complete
// (zlib rejects incomplete literal/length and code-length codes).  Returns false if no complete code was found (the
// caller falls back to the fixed code).
__device__ bool df_build_code(DfLds& L, const uint32_t* freq, int n, int limit, uint32_t* code, int lane) {
  uint32_t cnt = 0;
  for (int s = lane; s < n; s += WAVE) cnt += freq[s] != 0 ? 1u : 0u;
  const int used = (int)df_wave_sum(cnt);
  for (int s = lane; s < n; s += WAVE) {
    L.len[s] = 0;
    code[s] = 0;
    const uint32_t f = freq[s];
    if (!f) continue;
    uint32_t r = 0;
    for (int t = 0; t < n; t++) { const uint32_t g = freq[t]; r += (g != 0 && (g < f || (g == f && t < s))) ? 1u : 0u; }
    L.sorted[r] = (uint16_t)s;
  }
  df_sync();
  if (lane == 0) {
    bool ok = true;
    if (used == 0) { L.len[0] = 1; L.len[1] = 1; }
    else if (used == 1) { const int s0 = L.sorted[0]; L.len[s0] = 1; L.len[s0 == 0 ? 1 : 0] = 1; }
    else {
      const int Lf = used;
      for (int k = 0; k < Lf; k++) L.weight[k] = freq[L.sorted[k]];
      int i = 0, j = Lf;
      for (int nx = Lf; nx < 2 * Lf - 1; nx++) {
        int a, b;
        if (i < Lf && (j >= nx || L.weight[i] <= L.weight[j])) a = i++; else a = j++;
        if (i < Lf && (j >= nx || L.weight[i] <= L.weight[j])) b = i++; else b = j++;
        L.weight[nx] = L.weight[a] + L.weight[b];
        L.parent[a] = (uint16_t)nx; L.parent[b] = (uint16_t)nx;
      }
      bool over = false;
      L.weight[2 * Lf - 2] = 0;  // from here on weight[] of an internal node is its depth
      for (int k = 2 * Lf - 3; k >= 0; k--) {
        const uint32_t d = L.weight[L.parent[k]] + 1;
        if (k >= Lf) L.weight[k] = d;
        else { if (d > (uint32_t)limit) over = true; L.len[L.sorted[k]] = (uint8_t)(d > (uint32_t)limit ? (uint32_t)limit : d); }
      }
      if (over) {
        const uint32_t full = 1u << limit;
        uint32_t K = 0;
        for (int k = 0; k < Lf; k++) K += 1u << (limit - L.len[L.sorted[k]]);
        while (K > full) {
          int best = -1;
          uint32_t bl = 0;
          for (int k = 0; k < Lf; k++) { const uint32_t l = L.len[L.sorted[k]]; if (l < (uint32_t)limit && l > bl) { bl = l; best = k; } }
          if (best < 0) { ok = false; break; }
          L.len[L.sorted[best]]++;
          K -= 1u << (limit - bl - 1);
        }
        uint32_t slack = ok ? full - K : 0u;
        for (int l = limit; l >= 2 && slack; l--) {
          const uint32_t gain = 1u << (limit - l);
          for (int k = Lf - 1; k >= 0 && slack >= gain; k--)
            if (L.len[L.sorted[k]] == l) { L.len[L.sorted[k]] = (uint8_t)(l - 1); slack -= gain; }
        }
        if (slack) ok = false;
      }
    }
    if (ok) {
      uint32_t next[17];
      uint32_t blc[17];
      for (int l = 0; l <= 16; l++) blc[l] = 0;
      for (int s = 0; s < n; s++) blc[L.len[s]]++;
      blc[0] = 0;
      uint32_t c = 0;
      for (int l = 1; l <= 16; l++) { c = (c + blc[l - 1]) << 1; next[l] = c; }
      for (int s = 0; s < n; s++) {
        const uint32_t l = L.len[s];
        if (l) { const uint32_t cc = next[l]++; code[s] = ((__brev(cc) >> (32 - l)) << 8) | l; }
      }
    }
    L.ok = ok ? 1u : 0u;
  }
  df_sync();
  return L.ok != 0;
}

// the fixed code of RFC 1951 3.2.6 in the same tables
__device__ void df_fixed_code(DfLds& L, int lane) {
  for (int s = lane; s < DF_NLIT + 2; s += WAVE) {
    uint32_t c, l;
    if (s < 144) { c = 0x30u + s; l = 8; } else if (s < 256) { c = 0x190u + (s - 144); l = 9; } else if (s < 280) { c = s - 256; l = 7; } else { c = 0xC0u + (s - 280); l = 8; }
    L.cl[s] = ((__brev(c) >> (32 - l)) << 8) | l;
  }
  if (lane < DF_NDIST + 2) L.cd[lane] = ((__brev((uint32_t)lane) >> 27) << 8) | 5u;
  df_sync();
}

// serial bit writer of lane 0 into the (zeroed) staging window
__device__ __forceinline__ void df_put(uint32_t* W, uint32_t* bitpos, uint32_t v, uint32_t nb) {
  const uint32_t wi = *bitpos >> 5, lo = *bitpos & 31u;
  W[wi] |= v << lo;
  if (lo + nb > 32) W[wi + 1] |= v >> (32 - lo);
  *bitpos += nb;
}

// One BGZF member per wavefront.  Pass 1 parses (64 positions per step; a DF_WAYS-way 4096-entry hash table of 3-byte prefixes
// proposes its most recent candidates per lane, + the distance-1 candidate for runs; short matches far away are left to the
// literals (DF_FAR3 / DF_FAR4); lazy evaluation; a scalar walk over the lanes' token lengths picks the parse), stores the tokens and counts the symbols.  Then the block's own Huffman codes are built (df_build_code), the cost
// of the dynamic block (header included) is compared with the fixed code's, and pass 2 codes the tokens: bit offsets by a
// wave prefix sum, bits OR-ed into an LDS window that is flushed as whole dwords.  A member whose coded form would be
// larger than its payload is written as a stored block, so no member exceeds 64 KiB.
__global__ __launch_bounds__(WAVE) void k_bgzf_deflate(const uint8_t* __restrict__ payload, const uint64_t* __restrict__ m_off, uint32_t n_members,
                                                        const uint32_t* __restrict__ crc, uint8_t* __restrict__ slots, uint32_t slot_stride,
                                                        uint32_t* __restrict__ sizes, uint32_t* __restrict__ tokens_all) {
  __shared__ DfLds L;
  const uint32_t m = blockIdx.x;
  if (m >= n_members) return;
  const int lane = threadIdx.x;
  const uint8_t* in = payload + m_off[m];
  const uint32_t n = (uint32_t)(m_off[m + 1] - m_off[m]);
  uint8_t* slot = slots + (uint64_t)m * slot_stride;
  uint8_t* data = slot + 18;  // DEFLATE stream (dwords are stored unaligned: gfx950 global stores need no alignment)
  uint32_t* tokens = tokens_all + (uint64_t)m * 65536u;
  for (int k = lane; k < (1 << DF_HASH_BITS); k += WAVE) {
#pragma unroll
    for (int w = 0; w < DF_WAYS; w++) L.table[w][k] = (uint16_t)DF_NONE;
  }
  for (int k = lane; k < DF_NLIT + 2; k += WAVE) L.hl[k] = 0;
  if (lane < DF_NDIST + 2) L.hd[lane] = 0;
  if (lane < DF_NPRE + 1) L.hp[lane] = 0;
  for (int k = lane; k < DF_WORDS; k += WAVE) L.W[k] = 0;
  df_sync();
  // ---- pass 1: parse ----
  uint32_t pos = 0, ntok = 0, eb_acc = 0;
  while (pos < n) {
    const uint32_t p = pos + (uint32_t)lane;
    const bool inb = p < n;
    const bool can = p + 3 <= n;
    uint32_t v = 0;
    if (inb) v = ((const bw_u32*)(in + p))->v;  // (the payload buffer is padded)
    const uint32_t h = ((v & 0xFFFFFFu) * 0x9E3779B1u) >> (32 - DF_HASH_BITS);
    uint32_t cand[DF_WAYS];
#pragma unroll
    for (int w = 0; w < DF_WAYS; w++) cand[w] = can ? (uint32_t)L.table[w][h] : DF_NONE;
    df_sync();
    // the newest position moves in, the others move one way down (lanes with equal hashes: any of them is a valid candidate
    // for later positions)
    if (can) {
#pragma unroll
      for (int w = DF_WAYS - 1; w > 0; w--) L.table[w][h] = (uint16_t)cand[w - 1];
      L.table[0][h] = (uint16_t)p;
    }
    uint32_t best_len = 0, best_dist = 0;
    if (can) {
      const uint32_t cap = n - p < 258u ? n - p : 258u;
#pragma unroll
      for (int w = 0; w < DF_WAYS; w++) {
        const uint32_t c = cand[w];
        if (c != DF_NONE && c < p && p - c <= 32768u && best_len < cap) {
          // (zlib's shortcut -- skip a candidate that does not continue where the best match so far ends -- was measured and is
          // slower here, 10.1 -> 9.1 GB/s: the lanes run in lock step, so a wave pays for the full compare whenever one lane needs it)
          const uint32_t l = df_match_len(in + c, in + p, cap);
          const uint32_t d = p - c;
          if (l >= 3 && l > best_len && !(l == 3 && d > DF_FAR3) && !(l == 4 && d > DF_FAR4)) { best_len = l; best_dist = d; }
        }
      }
      if (p >= 1) {  // runs: the previous byte (positions of this very step are not in the table yet)
        const uint32_t l = df_match_len(in + p - 1, in + p, cap);
        if (l >= 3 && l > best_len) { best_len = l; best_dist = 1; }
      }
    }
    // lazy evaluation (zlib's): a match gives way to a literal when the next position has a longer one
    const uint32_t next_len = __shfl_down(best_len, 1, WAVE);
    if (best_len >= 3 && lane < WAVE - 1 && next_len > best_len) best_len = 0;
    const uint32_t tok_len = best_len >= 3 ? best_len : 1u;
    // greedy parse of the 64 positions: a scalar walk over the lanes' token lengths
    unsigned long long sel = 0;
    uint32_t k = 0;
    while (k < WAVE && pos + k < n) {
      sel |= 1ull << k;
      k += (uint32_t)__builtin_amdgcn_readlane((int)tok_len, (int)k);
    }
    const bool chosen = (sel >> lane) & 1ull;
    if (chosen) {
      const uint32_t idx = ntok + __builtin_amdgcn_mbcnt_hi((uint32_t)(sel >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sel, 0u));
      if (best_len >= 3) {
        uint32_t ls, leb, lex, ds, deb, dex;
        df_len_sym(best_len, &ls, &leb, &lex);
        df_dist_sym(best_dist, &ds, &deb, &dex);
        atomicAdd(&L.hl[ls], 1u);
        atomicAdd(&L.hd[ds], 1u);
        eb_acc += leb + deb;
        tokens[idx] = DF_TOK_MATCH | ((best_dist - 1u) << 8) | (best_len - 3u);
      } else {
        atomicAdd(&L.hl[v & 0xFFu], 1u);
        tokens[idx] = v & 0xFFu;
      }
    }
    ntok += (uint32_t)__popcll(sel);
    pos += k;
  }
  if (lane == 0) { tokens[ntok] = 256u; L.hl[256] = 1; }
  ntok++;
  const uint32_t ebits = df_wave_sum(eb_acc);
  df_sync();
  // ---- the block's own codes; cost of the dynamic block against the fixed one ----
  bool dynamic = df_build_code(L, L.hl, DF_NLIT, 15, L.cl, lane);
  dynamic = df_build_code(L, L.hd, DF_NDIST, 15, L.cd, lane) && dynamic;
  uint32_t hclen = 4, hlit = 0, hdist = 0, head_bits = 0;
  if (dynamic) {
    if (lane == 0) {
      int nl = DF_NLIT, nd = DF_NDIST;
      while (nl > 257 && (L.cl[nl - 1] & 0xFFu) == 0) nl--;
      while (nd > 1 && (L.cd[nd - 1] & 0xFFu) == 0) nd--;
      // code-length sequence with the zero runs folded (symbols 17 and 18)
      uint32_t nh = 0;
      const int tot = nl + nd;
      auto clen = [&](int q) { return q < nl ? (L.cl[q] & 0xFFu) : (L.cd[q - nl] & 0xFFu); };
      for (int q = 0; q < tot;) {
        const uint32_t c = clen(q);
        if (c == 0) {
          int run = 1;
          while (q + run < tot && run < 138 && clen(q + run) == 0) run++;
          if (run >= 11) { L.hsym[nh++] = (uint16_t)(18u | ((uint32_t)(run - 11) << 5) | (7u << 12)); L.hp[18]++; q += run; continue; }
          if (run >= 3) { L.hsym[nh++] = (uint16_t)(17u | ((uint32_t)(run - 3) << 5) | (3u << 12)); L.hp[17]++; q += run; continue; }
        }
        L.hsym[nh++] = (uint16_t)c;
        L.hp[c]++;
        q++;
      }
      L.nh = nh;
      L.weight[2 * DF_NLIT - 1] = (uint32_t)nl | ((uint32_t)nd << 16);
    }
    df_sync();
    dynamic = df_build_code(L, L.hp, DF_NPRE, 7, L.cp, lane);
    const uint32_t nlnd = L.weight[2 * DF_NLIT - 1];
    hlit = (nlnd & 0xFFFFu) - 257u;
    hdist = (nlnd >> 16) - 1u;
    if (dynamic) {
      const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      hclen = 19;
      while (hclen > 4 && (L.cp[order[hclen - 1]] & 0xFFu) == 0) hclen--;
      uint32_t hb = 0;
      for (uint32_t q = (uint32_t)lane; q < L.nh; q += WAVE) { const uint32_t e = L.hsym[q]; hb += (L.cp[e & 31u] & 0xFFu) + (e >> 12); }
      head_bits = 14u + 3u * hclen + df_wave_sum(hb);
    }
  }
  uint32_t dyn_bits = 0, fix_bits = 0;
  {
    uint32_t db = 0, fb = 0;
    for (int s = lane; s < DF_NLIT; s += WAVE) {
      const uint32_t f = L.hl[s];
      db += f * (L.cl[s] & 0xFFu);
      fb += f * (s < 144 ? 8u : s < 256 ? 9u : s < 280 ? 7u : 8u);
    }
    if (lane < DF_NDIST) { db += L.hd[lane] * (L.cd[lane] & 0xFFu); fb += L.hd[lane] * 5u; }
    dyn_bits = 3u + head_bits + df_wave_sum(db) + ebits;
    fix_bits = 3u + df_wave_sum(fb) + ebits;
  }
  if (!dynamic || fix_bits <= dyn_bits) { dynamic = false; df_fixed_code(L, lane); }
  // ---- block header ----
  uint32_t carry_bits = 0, word_base = 0;
  uint64_t total_bits = 0;
  if (lane == 0) {
    uint32_t bp = 0;
    if (!dynamic) df_put(L.W, &bp, 3u, 3);   // BFINAL = 1, BTYPE = 01
    else {
      const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      df_put(L.W, &bp, 5u, 3);               // BFINAL = 1, BTYPE = 10
      df_put(L.W, &bp, hlit, 5); df_put(L.W, &bp, hdist, 5); df_put(L.W, &bp, hclen - 4u, 4);
      for (uint32_t q = 0; q < hclen; q++) df_put(L.W, &bp, L.cp[order[q]] & 0xFFu, 3);
      for (uint32_t q = 0; q < L.nh; q++) {
        const uint32_t e = L.hsym[q], c = L.cp[e & 31u];
        df_put(L.W, &bp, c >> 8, c & 0xFFu);
        if (e >> 12) df_put(L.W, &bp, (e >> 5) & 0x7Fu, e >> 12);
      }
    }
    L.nh = bp;
  }
  df_sync();
  {
    const uint32_t T = L.nh;
    total_bits = T;
    const uint32_t full = T >> 5;
    for (uint32_t q = (uint32_t)lane; q < full; q += WAVE) bw_st32(data + 4ull * q, L.W[q]);
    const uint32_t carry_word = L.W[full];
    df_sync();
    for (int q = lane; q < DF_WORDS; q += WAVE) L.W[q] = (q == 0) ? carry_word : 0u;
    df_sync();
    word_base = full;
    carry_bits = T & 31u;
  }
  // ---- pass 2: code the tokens ----
  for (uint32_t base = 0; base < ntok; base += WAVE) {
    const uint32_t ti = base + (uint32_t)lane;
    uint64_t bits = 0;
    uint32_t nbits = 0;
    if (ti < ntok) {
      const uint32_t t = tokens[ti];
      if (t & DF_TOK_MATCH) {
        uint32_t ls, leb, lex, ds, deb, dex;
        df_len_sym((t & 0xFFu) + 3u, &ls, &leb, &lex);
        df_dist_sym(((t >> 8) & 0x7FFFu) + 1u, &ds, &deb, &dex);
        const uint32_t c1 = L.cl[ls], c2 = L.cd[ds];
        bits = (uint64_t)(c1 >> 8); nbits = c1 & 0xFFu;
        bits |= (uint64_t)lex << nbits; nbits += leb;
        bits |= (uint64_t)(c2 >> 8) << nbits; nbits += c2 & 0xFFu;
        bits |= (uint64_t)dex << nbits; nbits += deb;
      } else {
        const uint32_t c1 = L.cl[t & 0x1FFu];
        bits = (uint64_t)(c1 >> 8); nbits = c1 & 0xFFu;
      }
    }
    uint32_t tot;
    const uint32_t off = carry_bits + df_excl_scan(nbits, lane, &tot);
    if (nbits) {
      const uint32_t wi = off >> 5, lo = off & 31u;
      const uint64_t sh = bits << lo;               // <= 31 + 48 bits: three dwords at most
      atomicOr(&L.W[wi], (uint32_t)sh);
      if (lo + nbits > 32) atomicOr(&L.W[wi + 1], (uint32_t)(sh >> 32));
      if (lo + nbits > 64) atomicOr(&L.W[wi + 2], (uint32_t)(bits >> (64 - lo)));
    }
    df_sync();
    const uint32_t T = carry_bits + tot;
    const uint32_t full = T >> 5;   // whole dwords: flushed; the partial one is carried into the next step
    for (uint32_t q = (uint32_t)lane; q < full; q += WAVE) bw_st32(data + 4ull * (word_base + q), L.W[q]);
    const uint32_t carry_word = L.W[full];
    df_sync();
    for (uint32_t q = (uint32_t)lane; q <= full + 1 && q < (uint32_t)DF_WORDS; q += WAVE) L.W[q] = (q == 0) ? carry_word : 0u;
    df_sync();
    word_base += full;
    carry_bits = T & 31u;
    total_bits += tot;
  }
  if (lane == 0 && carry_bits) bw_st32(data + 4ull * word_base, L.W[0]);  // the last partial dword
  df_sync();
  uint32_t data_bytes = (uint32_t)((total_bits + 7) >> 3);
  if (data_bytes > n + 5u) {
    // the coded form is larger than the payload: a stored block instead (BFINAL = 1, BTYPE = 00, LEN, NLEN, bytes),
    // so that no member exceeds 64 KiB (65280 + 5 + 26 bytes)
    if (lane == 0) { data[0] = 1; data[1] = (uint8_t)n; data[2] = (uint8_t)(n >> 8); data[3] = (uint8_t)~n; data[4] = (uint8_t)(~n >> 8); }
    for (uint32_t q = lane; q < n; q += WAVE) data[5 + q] = in[q];
    data_bytes = n + 5u;
    df_sync();
  }
  // gzip member header with the BGZF extra subfield, and the trailer (SAM spec 4.1)
  const uint32_t bsize = 18u + data_bytes + 8u;
  if (lane == 0) {
    const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    for (int q = 0; q < 16; q++) slot[q] = hdr[q];
    slot[16] = (uint8_t)(bsize - 1); slot[17] = (uint8_t)((bsize - 1) >> 8);
    uint8_t* t = data + data_bytes;
    const uint32_t cr = crc[m];
    t[0] = (uint8_t)cr; t[1] = (uint8_t)(cr >> 8); t[2] = (uint8_t)(cr >> 16); t[3] = (uint8_t)(cr >> 24);
    t[4] = (uint8_t)n; t[5] = (uint8_t)(n >> 8); t[6] = (uint8_t)(n >> 16); t[7] = (uint8_t)(n >> 24);
    sizes[m] = bsize;
  }
}

// members back to back: out[off[m] ..] = slot m
__global__ __launch_bounds__(256) void k_compact_members(const uint8_t* __restrict__ slots, uint32_t slot_stride, const uint32_t* __restrict__ sizes,
                                                          const uint64_t* __restrict__ off, uint8_t* __restrict__ out) {
  const uint32_t m = blockIdx.x;
  const uint8_t* s = slots + (uint64_t)m * slot_stride;
  uint8_t* d = out + off[m];
  const uint32_t n = sizes[m];
  for (uint32_t k = threadIdx.x * 4; k + 4 <= n; k += 256 * 4) bw_st32(d + k, ((const bw_u32*)(s + k))->v);
  if (threadIdx.x < (n & 3u)) d[(n & ~3u) + threadIdx.x] = s[(n & ~3u) + threadIdx.x];
}

void launch_bgzf_deflate(const uint8_t* payload, const uint64_t* m_off, uint32_t n_members, const uint32_t* crc, uint8_t* slots,
                         uint32_t slot_stride, uint32_t* sizes, uint32_t* tokens, hipStream_t st) {
  if (!n_members) return;
  hipLaunchKernelGGL(k_bgzf_deflate, dim3(n_members), dim3(WAVE), 0, st, payload, m_off, n_members, crc, slots, slot_stride, sizes, tokens);
}
void launch_compact_members(const uint8_t* slots, uint32_t slot_stride, const uint32_t* sizes, const uint64_t* off, uint32_t n_members,
                            uint8_t* out, hipStream_t st) {
  if (!n_members) return;
  hipLaunchKernelGGL(k_compact_members, dim3(n_members), dim3(256), 0, st, slots, slot_stride, sizes, off, out);
}

}  // namespace bioscan
