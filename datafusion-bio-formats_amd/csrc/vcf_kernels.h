// vcf_kernels.h -- launch wrappers of the VCF text-path kernels (vcf_kernels.hip).  Internal header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bioscan.h"

namespace bioscan {

enum VcfDevError : uint32_t {
  VERR_NONE = 0, VERR_BLANK_LINE = 1, VERR_SHORT_RECORD = 2, VERR_BAD_POS = 3, VERR_MISSING_START = 4, VERR_BAD_END = 5,
  VERR_BAD_QUAL = 6, VERR_FLOAT_PRECISION = 7, VERR_DUP_INFO_KEY = 8, VERR_BAD_INT = 9, VERR_BAD_FLOAT = 10,
  VERR_INVALID_FLAG = 11, VERR_PERCENT = 12, VERR_BAD_GT = 13, VERR_BAD_CHAR = 14, VERR_UNSUPPORTED_INFO = 15
};

// ---- header types of INFO / FORMAT keys ------------------------------------------------------------------------
// noodles types every entry of INFO (`info.iter(header)`, physical_exec.rs:561-571) and every value of a selected sample
// (`sample.iter(header)`, :1661-1666) by the header as it walks them, whether or not the scan has a column for the key: a scalar
// that does not parse is the record's error.  Lists and genotypes stay lazy and are only walked for keys the table has a builder
// for (:572-611, :1668-1760).  The kernels therefore look every key they do NOT extract up in a table of the header's
// declarations and check its value the same way.
enum VcfCheckKind : uint32_t {
  CK_NONE = 0,         // nothing to check (a lazy list / genotype of a key without a builder)
  CK_INT = 1, CK_FLOAT = 2, CK_STR = 3, CK_FLAG = 4, CK_CHAR = 5,
  CK_UNSUPPORTED = 6,  // INFO Character key with a builder: any value is "Unsupported INFO value type" (:632-636)
  CK_GT = 7,           // genotype of a key with a builder
  CK_LIST = 8          // bit: a list whose elements (of the scalar kind in the low bits) are walked
};
// key8 = the key's first (up to) eight bytes, little endian, zero padded: most keys are compared and hashed as one 64-bit word;
// the bytes of a longer key are at keys + (off & 0xFFFF).  len_kind = length | (kind + 1) << 24; 0 = empty slot.  off >> 16 = 1 + the
// index of the key among the keys the scan extracts (0: not one of them) -- k_vcf_info_locate finds a selected key with the same
// lookup that types an unselected one (has_sel; r04: comparing every entry with each of the K selected keys was a third of the kernel).
struct VcfTypeSlot { uint64_t key8; uint32_t off, len_kind; };
struct VcfTypeTable {
  const uint8_t* keys;
  const VcfTypeSlot* slots;   // open addressing, linear probing, mask + 1 slots (a power of two, at most half full)
  uint32_t mask;
  uint32_t miss_kind;         // a key the header does not declare: String, Number=1 (noodles' default)
  uint32_t has_sel;           // the slots carry selected-key indices
};
__host__ __device__ inline uint32_t vcf_key_hash(uint64_t key8, uint32_t n) {
  return (uint32_t)(((key8 ^ n) * 0x9E3779B97F4A7C15ull) >> 40);
}

// delimiter index over u[lo, hi): positions of '\n' and '\t', and for each newline the number of tabs before it
uint64_t vcf_delim_chunks(uint64_t lo, uint64_t hi);
void launch_vcf_delim_count(const uint8_t* u, uint64_t lo, uint64_t hi, uint32_t* cnt_nl, uint32_t* cnt_tab, hipStream_t st);
void launch_vcf_delim_write(const uint8_t* u, uint64_t lo, uint64_t hi, const uint64_t* base_nl, const uint64_t* base_tab,
                            uint64_t* nl, uint64_t* nl_tabs, uint64_t* tab, hipStream_t st);

struct VcfLines {
  const uint64_t* nl;       // [n_nl] newline positions >= x0
  const uint64_t* nl_tabs;  // [n_nl] tabs in [x0, nl[j])
  const uint64_t* tab;      // [n_tab]
  uint64_t n_nl, n_tab;
  uint64_t x0, hi;          // first line start, end of the decoded bytes
  uint64_t n_lines;         // n_nl (+1 for an unterminated last line at the end of the data)
};
// need_end: 0 no END wanted; else 1 + the check kind of the key END (CK_INT: its value is the end; anything else: checked
// like any other entry, the end stays POS + len(REF) - 1), | 0x100 when single-base substitutions are walked too (indexed
// scans).  T: the INFO declarations as `Info::get` sees them -- no key has a builder -- (entries in front of END are typed).
void launch_vcf_keys(const uint8_t* u, VcfLines L, uint32_t* pos, uint32_t* vend, uint8_t* flags, int need_end, VcfTypeTable T,
                     uint32_t* err, hipStream_t st);

struct VcfFilterTerm {
  int32_t field;     // 0 chrom, 1 start, 2 end, 3 id
  int32_t op;        // bioscan_filter_op
  int32_t n_vals;
  int32_t has_null;
  int32_t more;      // [NOT] IN only: the next term holds further literals of the same list
  int32_t pad_;
  double vals[8];
  uint32_t str_off[8], str_len[8];  // into the string blob (string fields)
};
struct VcfRowSelect {
  int32_t mode;              // 0 every line, 1 region query
  int32_t n_chunks;
  uint64_t i_lo, i_hi;       // line range examined
  uint32_t chrom_off, chrom_len;  // region chrom in the string blob
  int64_t q_start1, q_end1;  // noodles interval (1-based inclusive)
  int64_t start1, end1;      // reference's start filter; <= 0 = unbounded
  int32_t zero_based;
  int32_t n_terms;
};
void launch_vcf_row_flags(const uint8_t* u, VcfLines L, const uint32_t* pos, const uint32_t* vend, const uint8_t* flags,
                          VcfRowSelect S, const uint64_t* chunks, const VcfFilterTerm* terms, const uint8_t* strs, uint32_t* keep,
                          hipStream_t st);
void launch_vcf_compact(const uint32_t* keep, const uint64_t* scan, uint64_t n, uint64_t i_lo, uint64_t* rows, uint64_t row_base,
                        uint64_t cap, hipStream_t st);
void launch_vcf_line_lower_bound(VcfLines L, uint64_t off, unsigned long long* result, hipStream_t st);
void launch_vcf_iota_rows(uint64_t* rows, uint64_t n, hipStream_t st);

struct VcfCoreCols {  // nullptr = not projected
  uint64_t* src_chrom; uint32_t* len_chrom;
  uint32_t* start; uint32_t* end;
  uint64_t* src_id; uint32_t* len_id;
  uint64_t* src_ref; uint32_t* len_ref;
  uint64_t* src_alt; uint32_t* len_alt;
  double* qual; uint64_t* v_qual;
  uint64_t* src_filter; uint32_t* len_filter;
};
void launch_vcf_core(const uint8_t* u, VcfLines L, const uint64_t* rows, uint64_t n, const uint32_t* pos, const uint32_t* vend,
                     const uint8_t* flags, VcfCoreCols C, int zero_based, uint32_t* err, hipStream_t st);
void launch_replace_byte(uint8_t* d, uint64_t n, uint8_t from, uint8_t to, hipStream_t st);

// key_unsupported[k] != 0: selected key k is a Character key (a value is an error).  T: the INFO declarations, for every
// entry whose key is not one of the K selected ones.
void launch_vcf_info_locate(const uint8_t* u, VcfLines L, const uint64_t* rows, uint64_t n, const uint8_t* keys, const uint32_t* key_off,
                            const uint8_t* key_unsupported, int K, VcfTypeTable T, uint64_t* sp_off, uint32_t* sp_len,
                            uint8_t* sp_state, uint32_t* err, hipStream_t st);

// typed span kernels: span c = u[off[c], off[c]+len[c]) with state[c] (0 absent, 1 value, 2 bare key)
void launch_span_num(const uint8_t* u, const uint64_t* off, const uint32_t* len, const uint8_t* state, uint64_t N, int kind,
                     uint32_t* values, uint64_t* valid, uint32_t* err, hipStream_t st);
void launch_span_flag(const uint8_t* u, const uint64_t* off, const uint32_t* len, const uint8_t* state, uint64_t N, uint64_t* bits,
                      uint32_t* err, hipStream_t st);
// out_len = length after percent-decoding; *pct_flag |= 1 when some value holds an escape (the scatter then decodes)
void launch_span_str(const uint8_t* u, const uint64_t* off, const uint32_t* len, const uint8_t* state, uint64_t N, uint32_t* out_len,
                     uint64_t* valid, uint32_t* pct_flag, uint32_t* err, hipStream_t st);
void launch_scatter_pct(const uint8_t* u, const uint64_t* src, uint64_t n, const uint64_t* off64, uint8_t* dst, hipStream_t st);
void launch_span_list_count(const uint8_t* u, const uint64_t* off, const uint32_t* len, const uint8_t* state, uint64_t N, uint32_t* cnt,
                            uint64_t* valid, hipStream_t st);
void launch_span_list_elems(const uint8_t* u, const uint64_t* off, const uint32_t* len, const uint8_t* state, uint64_t N,
                            const uint64_t* eoff, int kind, uint32_t* values, uint64_t* esrc, uint32_t* elen, uint8_t* evalid,
                            uint32_t* pct_flag, uint32_t* err, hipStream_t st);
void launch_pack_bits(const uint8_t* bytes, uint64_t n, uint64_t* words, hipStream_t st);
void launch_stride_offsets(uint64_t* off, uint64_t n, uint64_t stride, hipStream_t st);
void launch_off64_to_32(const uint64_t* off, uint64_t n, int32_t* out, hipStream_t st);   // (values below 2^31)

// cmap[r]: nibble j < 15 = check kind (low 3 bits; bit 3 = list) of the row's FORMAT key j when that key is NOT one of the S
// selected ones (a selected key: CK_CHAR when char_mask has its bit -- a Character scalar, extracted as a string --, else 0);
// nibble 15 != 0: the row has more keys than that.  err[3] is raised when some cmap of the chunk is not 0: only then
// launch_vcf_format_check has anything to do.
void launch_vcf_format_keys(const uint8_t* u, VcfLines L, const uint64_t* rows, uint64_t n, const uint8_t* keys, const uint32_t* key_off,
                            int S, VcfTypeTable T, uint32_t char_mask, int16_t* fpos, uint64_t* cmap, uint32_t* err, hipStream_t st);
void launch_vcf_format_check(const uint8_t* u, VcfLines L, const uint64_t* rows, uint64_t n, const int32_t* sample_col, int NS,
                             const uint64_t* cmap, VcfTypeTable T, uint32_t* err, hipStream_t st);
// FORMAT keys parsed inside the cell kernel: kind 1 Int32 / 2 Float32 -> values[s][c] + validity words valid[s];
// kind 3 GT -> values[s][c] = rendered length, src[s][c] = its first byte (offset into u) + validity; kind 0 = emit
// a span for the typed span kernels (other strings, lists).  Only the first VCF_MAX_DIRECT selected keys can be direct.
constexpr int VCF_MAX_DIRECT = 8;
struct VcfCellDirect {
  int32_t kind[VCF_MAX_DIRECT];
  uint32_t* values[VCF_MAX_DIRECT];
  uint64_t* valid[VCF_MAX_DIRECT];
  uint64_t* src[VCF_MAX_DIRECT];
};
// The error word of a scan is followed by a queue of float cells whose rounding the parsing kernels could not prove:
// err[0] error code, err[1] number of queued cells, err[2] "a GT cell has to be rendered again" (launch_gt_render), err[3] "a FORMAT
// value is checked without being extracted" (launch_vcf_format_check), entries from err + 4.  launch_f32_fix rewrites them exactly.
struct F32Fix { const uint8_t* p; void* dst; uint32_t len, as_f64, approx, pad; };
constexpr uint32_t F32_FIX_CAP = 65536;   // 2 MB of queue per scan; more such cells in one chunk are an error
constexpr size_t VCF_ERR_DWORDS = 4 + F32_FIX_CAP * (sizeof(F32Fix) / 4);
void launch_f32_fix(uint32_t* err, uint32_t n_queued, hipStream_t st);
void launch_gt_render(const uint8_t* u, const uint64_t* src, const uint64_t* off, uint8_t* values, uint64_t N, hipStream_t st);
void launch_vcf_format_cells(const uint8_t* u, VcfLines L, const uint64_t* rows, uint64_t n, const int32_t* sample_col, int NS,
                             const int16_t* fpos, int S, int gt_field, VcfCellDirect D, uint64_t* sp_off, uint32_t* sp_len,
                             uint8_t* sp_state, uint32_t* err, hipStream_t st);

// list UDFs: off = u64 list offsets (n+1), values = 32-bit elements, evalid / lvalid = validity words or nullptr
void launch_list_avg(const uint64_t* off, const uint32_t* values, const uint64_t* evalid, const uint64_t* lvalid, uint64_t n,
                     int is_float, double* out, uint8_t* out_valid, hipStream_t st);
// list_and: bit-packed Boolean values / validity of both lists, u64 offsets; out_val / out_valid zero-initialised
void launch_list_and(const uint64_t* off_l, const uint64_t* off_r, const uint64_t* off_o, const uint64_t* lval, const uint64_t* lvalid,
                     const uint64_t* rval, const uint64_t* rvalid, uint64_t n, uint64_t* out_val, uint64_t* out_valid, hipStream_t st);
// vcf_set_gts: per GT element -> source offset (into [GT bytes | replacement]), length, validity byte
void launch_set_gts_plan(const uint64_t* off_g, const uint64_t* goff, const uint64_t* gvalid, const uint64_t* off_m,
                         const uint64_t* mlvalid, const uint64_t* mval, const uint64_t* mvalid, uint64_t n, uint64_t rep_off,
                         uint32_t rep_len, uint64_t* src, uint32_t* len, uint8_t* ovalid, hipStream_t st);
void launch_count_bits(const uint64_t* bits, const uint64_t* valid, uint64_t n_elems, unsigned long long* counts, hipStream_t st);
// vcf_an / vcf_ac / vcf_af: per row (list of GT strings: element byte offsets goff, element validity words or null, row offsets
// off_g) the number of called alleles and the largest allele index; then the per-ALT-allele counts into zeroed `counts`
void launch_gt_stats(const uint8_t* bytes, const uint64_t* goff, const uint64_t* gvalid, const uint64_t* off_g, uint64_t n, int32_t* an,
                     unsigned long long* max_allele, hipStream_t st);
void launch_gt_ac(const uint8_t* bytes, const uint64_t* goff, const uint64_t* gvalid, const uint64_t* off_g, uint64_t n, const uint64_t* out_off,
                  int32_t* counts, hipStream_t st);
void launch_list_cmp(const uint32_t* values, uint64_t n_elems, int is_float, int op, uint32_t thr_bits, uint64_t* out_bits, hipStream_t st);

}  // namespace bioscan
