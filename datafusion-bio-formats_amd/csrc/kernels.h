// kernels.h -- launch wrappers of the gfx950 kernels (device code lives in kernels.hip).
// Internal header: the public boundary is include/bioscan.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bioscan {

constexpr uint64_t SEG_NONE = ~0ull;       // segment has no record start
constexpr uint64_t SEG_BAD = ~0ull - 1;    // walk ran into an impossible block_size
constexpr uint64_t SEG_PARTIAL = 1ull << 62;  // | start of a record that runs past the end of the chunk (see k_seg_walk)
constexpr uint32_t SEG_BYTES = 65536;      // record-chain segment size (bytes of inflated stream)

// per-block inflate status codes (0 = ok)
enum InflateStatus : uint32_t {
  INF_OK = 0, INF_BAD_HEADER = 1, INF_BAD_BTYPE = 2, INF_BAD_CODE = 3, INF_BAD_DIST = 4,
  INF_OVERRUN = 5, INF_SIZE_MISMATCH = 6, INF_BAD_STORED = 7, INF_CRC_MISMATCH = 8,
  INF_RETRY = 9   // K1 v4: the Huffman codes of a block need more sub-table space than its LDS pool holds; the wide-table kernel decodes the member
};

// ---- K1: BGZF inflate -------------------------------------------------------------------------
// comp: compressed file bytes (padded by >= 1 KiB readable slack), blk_coff[i] = byte offset of
// BGZF member i, blk_uoff[i] = offset of its payload in `out`; blk_uoff[n] = total.
// K1 (inflate_v3.hip): long sub-streams, checkpointed count passes, segment mini-rounds.  Per-wave scratch stride in
// u64: the match list of one mini-round (V3_ML_ENTRIES) followed by the checkpoint rows of one round.
constexpr uint32_t V3_ML_ENTRIES = 1536;
#ifndef V3_CK_MAX
#define V3_CK_MAX 24          // checkpoints per lane and pass; a pass that needs more restarts its round with short sub-streams
#endif
constexpr uint32_t V3_CK_DWORDS = V3_CK_MAX * 3 * 64;  // V3_CK_MAX rows x (pos, acc, state) x 64 lanes
constexpr uint32_t V3_SCRATCH_STRIDE = V3_ML_ENTRIES + V3_CK_DWORDS / 2;
int v3_resident_wg_per_cu();
void launch_bgzf_inflate_v3(const uint8_t* comp, const uint64_t* blk_coff, const uint64_t* blk_uoff, uint8_t* out,
                            uint32_t n_blocks, uint32_t* status, uint32_t* counter, unsigned long long* scratch,
                            uint32_t grid, uint32_t* dbg, hipStream_t st, uint32_t* slots = nullptr, uint32_t n_slots = 0,
                            uint32_t per_wave = 0, uint32_t wpw = 0, const uint32_t* pre = nullptr, bool retry_only = false);
// K1 with the v4 decode step (inflate_v4.hip): same launch contract and scratch layout.
int v4_resident_wg_per_cu();
void launch_bgzf_inflate_v4(const uint8_t* comp, const uint64_t* blk_coff, const uint64_t* blk_uoff, uint8_t* out,
                            uint32_t n_blocks, uint32_t* status, uint32_t* counter, unsigned long long* scratch,
                            uint32_t grid, uint32_t* dbg, hipStream_t st, uint32_t* slots = nullptr, uint32_t n_slots = 0,
                            uint32_t per_wave = 0, uint32_t wpw = 0, const uint32_t* pre = nullptr);
// K0 (inflate_headers.hip): the first DEFLATE block header of every member, one member per lane; rec holds V3_PRE_DWORDS
// dwords per member (flags, header bits, 320 code lengths as nibbles) that K1 reads through its `pre` argument.
constexpr uint32_t V3_PRE_DWORDS = 42;
void launch_bgzf_headers(const uint8_t* comp, const uint64_t* blk_coff, uint32_t n_blocks, uint32_t* rec, hipStream_t st);
// K2: CRC32 of each inflated block vs the BGZF trailer (validation mode).
// nl_cnt != nullptr (FASTQ): also adds the newlines of every 16 KiB tile of the text buffer to nl_cnt[tile] (zeroed by the caller);
// a member byte at inflated offset o lies at buffer position o + nl_bias, and the buffer starts on a 16-byte boundary.
void launch_bgzf_crc32(const uint8_t* comp, const uint64_t* blk_coff, const uint64_t* blk_uoff,
                       const uint8_t* out, uint32_t n_blocks, uint32_t* status, hipStream_t st, uint32_t* nl_cnt = nullptr,
                       uint64_t nl_bias = 0);

// ---- scans ------------------------------------------------------------------------------------
// out[0..n] exclusive prefix sums of in[0..n) (out has n+1 entries); tmp must hold
// scan_tmp_elems(n) uint64.
size_t scan_tmp_elems(uint64_t n);
void launch_exclusive_scan_u32_to_u64(const uint32_t* in, uint64_t* out, uint64_t n, uint64_t* tmp, hipStream_t st);

// ---- K3: record boundary scan -----------------------------------------------------------------
struct ChainBuffers {
  uint64_t* entry;   // [nseg]
  uint64_t* exit_;   // [nseg]
  uint32_t* count;   // [nseg]
  uint32_t* dirty;   // [nseg]
  uint32_t* nfix;    // [1] device counter
  uint32_t* err;     // [1] device error flag
  uint32_t* starts;  // [nseg][SEG_SLOT] record starts found by the walk, segment-relative
};
constexpr uint32_t SEG_SLOT = 2048;  // a record is >= 36 bytes: at most 1820 starts per 64 KiB segment
void launch_seg_guess(const uint8_t* u, uint64_t ulen, uint64_t first_rec, uint64_t nseg, int32_t n_ref,
                      ChainBuffers cb, hipStream_t st);
void launch_seg_walk(const uint8_t* u, uint64_t ulen, uint64_t nseg, ChainBuffers cb, int only_dirty, int allow_partial, hipStream_t st);
void launch_seg_verify(uint64_t ulen, uint64_t first_rec, uint64_t nseg, ChainBuffers cb, hipStream_t st);
void launch_last_exit(const uint64_t* exit_, uint64_t nseg, unsigned long long* res, hipStream_t st);
void launch_first_bad_status(const uint32_t* status, uint32_t n, uint32_t* res, hipStream_t st);
void launch_seg_gather(uint64_t nseg, ChainBuffers cb, const uint64_t* base, uint64_t* rec_off, hipStream_t st);

// ---- record key table (refid,pos,end,flag,mapq per record) ------------------------------------
struct RecKeys {
  int32_t* refid;
  int32_t* pos;      // 0-based, -1 = none
  int32_t* end1;     // 1-based inclusive end, 0 = None
  uint32_t* flag_mapq;  // flag | mapq << 16
};
void launch_rec_keys(const uint8_t* u, const uint64_t* rec_off, uint64_t n, RecKeys k, uint32_t* err, hipStream_t st);  // err = 8: invalid record

// ---- row selection ----------------------------------------------------------------------------
// Residual filter program: conjunction of terms on numeric fields / chrom index.
struct FilterTerm {
  int32_t field;     // 0 chrom(refid), 1 start, 2 end, 3 mapping_quality, 4 flags
  int32_t op;        // bioscan_filter_op
  int32_t n_vals;    // number of literal values in this term (<= 8)
  int32_t has_null;  // in-list contained NULL / non-numeric literal
  int32_t more;      // [NOT] IN only: the next term holds further literals of the same list (lists of any length)
  int32_t pad_;
  double vals[8];    // numeric literals, or ref index for chrom (-1 = name not in header)
};
struct RowSelect {
  int32_t mode;          // 0 all records, 1 mapped region, 2 unmapped tail of ref, 3 no-coor
  int32_t ref;           // region reference index
  int64_t start1, end1;  // region bounds 1-based inclusive, <=0 / INT64_MAX = unbounded
  int64_t q_start1;      // noodles interval start used by intersects()
  uint64_t i_lo, i_hi;   // record index range considered (mode 2: run bounds)
  int32_t zero_based;
  int32_t n_terms;
  // mode 1: the region's merged BAI chunks as [begin, end) pairs of absolute inflated offsets, entries ch_lo .. ch_lo + ch_n of
  // the item's chunk table; a record belongs to the answer only if it STARTS inside one of them (the reference reads the
  // chunks, not what lies between them: a record the index does not list is not returned).  ch_n == 0: no such test.
  uint32_t ch_lo, ch_n;
};
void launch_row_flags(RecKeys k, uint64_t n, RowSelect sel, const FilterTerm* terms_dev, uint32_t* keep, int accumulate, hipStream_t st);
void launch_row_flags_rec(const uint8_t* u, const uint64_t* rec_off, uint64_t n, const RowSelect* sels_dev, int n_sel, const FilterTerm* terms_dev,
                          uint32_t* keep, uint32_t* err, hipStream_t st, const uint64_t* chunk_tab = nullptr, uint64_t buf_base_abs = 0);
void launch_compact_rows(const uint64_t* rec_off, const uint32_t* keep, const uint64_t* keep_scan, uint64_t n,
                         uint64_t* rows, uint64_t row_base, hipStream_t st);
// tail run finder: first index >= i0 with refid==ref  /  first index > a with refid != ref
void launch_find_first(const int32_t* refid, uint64_t n, uint64_t from, int32_t ref, int want_equal,
                       unsigned long long* result, hipStream_t st);
void launch_lower_bound_u64(const uint64_t* arr, uint64_t n, uint64_t key, unsigned long long* result, hipStream_t st);

// ---- per-batch offsets of columns that keep an offset array (tags, wide qualities) ------------
void launch_batch_offsets(const uint64_t* off64, uint64_t n_rows, uint32_t batch_size, uint32_t phase, int32_t* off32, hipStream_t st);
void launch_batch_bases(const uint64_t* off64, uint64_t nb, uint32_t bs, uint32_t phase, uint64_t* base, hipStream_t st);
// exact path for quality bytes >= 95 (`char::from(q + 33)` is a two-byte UTF-8 char): own length pass + scatter
void launch_qual_wide_len(const uint8_t* u, const uint64_t* rows, uint64_t row0, uint64_t n, uint32_t* len_qual, hipStream_t st);
void launch_qual_wide_scatter(const uint8_t* u, const uint64_t* rows, uint64_t n, const uint64_t* off64, uint8_t* dst, hipStream_t st);

// ---- bam_rows.hip: the twelve core columns in two passes (fixed columns + tile sums; offsets + scatter) ----------
constexpr int ROWS_TILE = 256;
struct RowsCols {   // device pointers; nullptr = not projected
  uint32_t* start; uint32_t* end; uint32_t* flags; uint32_t* mapq; uint32_t* mate_start; int32_t* tlen;
  uint64_t* v_chrom; uint64_t* v_start; uint64_t* v_end; uint64_t* v_mate_chrom; uint64_t* v_mate_start;  // validity words
  // variable-length columns k: 0 name, 1 chrom, 2 cigar, 3 mate_chrom, 4 sequence, 5 quality_scores
  uint8_t* val[6];     // values (pass 2)
  int32_t* off32[6];   // per-batch int32 offsets [n_batches][batch_size + 1] (pass 2)
  uint64_t* base[6];   // first byte of every batch [n_batches] (pass 2)
  uint32_t want;       // bit k: column k is projected
};
// pass 1 + the scan of its tile sums: tile_sums holds bam_rows_scratch_elems(n) u64 (6 x (n_tiles + 1) tile sums, then the
// group totals of the two-level scan); afterwards entry [k][n_tiles] is column k's total
size_t bam_rows_scratch_elems(uint64_t n);
void launch_bam_rows_pass1(const uint8_t* u, const uint64_t* rows, uint64_t n, RowsCols c, const uint32_t* ref_name_len, int32_t n_ref,
                           int32_t zero_based, int32_t binary_cigar, uint64_t* tile_sums, uint32_t* err, hipStream_t st);
void launch_bam_rows_pass2(const uint8_t* u, const uint64_t* rows, uint64_t n, RowsCols c, const uint8_t* ref_names,
                           const uint32_t* ref_name_off, const uint32_t* ref_name_len, int32_t n_ref, int32_t binary_cigar,
                           uint32_t batch_size, uint32_t phase, const uint64_t* tile_sums, uint32_t* qual_wide, hipStream_t st);

// ---- bam_write.hip: Arrow columns -> BAM records -> BGZF members (the write side of the path) -----------------
struct SerCols {   // one RecordBatch on the device, Arrow layouts as uploaded; `offset` = the arrays' logical offset
  int64_t offset;
  const int32_t* name_off; const uint8_t* name; const uint8_t* name_valid;
  const int32_t* cigar_off; const uint8_t* cigar; int32_t cigar_binary;
  const int32_t* seq_off; const uint8_t* seq;
  const int32_t* qual_off; const uint8_t* qual;
  const uint32_t* start; const uint8_t* start_valid;
  const uint32_t* flags; const uint32_t* mapq;
  const uint32_t* mate_start; const uint8_t* mate_start_valid;
  const int32_t* tlen;
  const int32_t* refid; const int32_t* mate_refid;  // per row (index 0 = row 0 of the batch), from chrom / mate_chrom; -1 = none
  int32_t zero_based;
};
// One tag column of the batch (build_tag_data, bio-format-core/src/sam_tag_io.rs:109-147): a non-NULL value becomes the aux
// field tag[2] + type + value.  kind / ekind: Arrow storage of the column / of a list's elements.
enum SerKind : uint8_t { SK_I8 = 1, SK_I16, SK_I32, SK_I64, SK_U8, SK_U16, SK_U32, SK_U64, SK_F32, SK_F64, SK_UTF8, SK_LIST };
struct SerTagCol {
  uint8_t tag[2];
  uint8_t sam_type;        // 'i' 'c' 's' 'C' 'S' 'I' 'f' 'Z' 'H' 'A' 'B'
  uint8_t subtype;         // 'B': element type on disk ('c' 'C' 's' 'S' 'i' 'I' 'f')
  uint8_t kind, ekind;
  int64_t offset, eoffset; // logical offsets of the column / of the list's child array
  const uint8_t* valid;    // validity bitmap or null
  const uint8_t* values;   // fixed width: elements; Utf8: bytes; List: the child's elements
  const int32_t* off;      // Utf8 / List offsets
  const uint8_t* evalid;   // the list child's validity bitmap or null
};
struct SerTags { int32_t n; const SerTagCol* cols; };
// tag_err (atomicMin, start at ~0): row << 16 | column << 8 | code -- 20 integer does not fit its SAM type, 21 float does not fit 'f',
// 22 invalid hex string, 23 character tag is not one ASCII byte, 24 character tag value does not fit a byte, 25 NULL element in an
// array tag, 26 array element does not fit its subtype.  The host words the message from the batch it still holds.
// err: 1 flag > 65535, 2 malformed CIGAR, 3 quality / sequence length mismatch, 4 read name too long, 5 more than 65535 CIGAR ops
void launch_ser_sizes(SerCols c, SerTags t, uint64_t n, uint32_t* rec_bytes, uint32_t* err, unsigned long long* tag_err, hipStream_t st);
void launch_ser_write(SerCols c, SerTags t, uint64_t n, const uint64_t* rec_off, uint8_t* out, uint32_t* err, hipStream_t st);
// crc[b] = CRC32 of payload[off[b] .. off[b + 1]) (k_bgzf_crc32 in store mode)
void launch_crc32_store(const uint8_t* payload, const uint64_t* off, uint32_t n_members, uint32_t* crc, hipStream_t st);
// one BGZF member per payload range: complete members (header, DEFLATE data, CRC32, ISIZE) in slots of `slot_stride` bytes
// (>= BGZF_SLOT_BYTES), their sizes in sizes[]; launch_compact_members lays them back to back at out + off[m]
constexpr uint32_t BGZF_MAX_PAYLOAD = 65280;   // what noodles-bgzf / htslib put into one member
constexpr uint32_t BGZF_SLOT_BYTES = 81920;    // coded worst case (a member that would exceed its payload + 5 is stored; 9 bits per byte while coding) + header, trailer, slack
// tokens: scratch of BGZF_TOKENS_PER_MEMBER dwords per member (the parse of pass 1, read back by pass 2)
constexpr uint32_t BGZF_TOKENS_PER_MEMBER = 65536;
void launch_bgzf_deflate(const uint8_t* payload, const uint64_t* m_off, uint32_t n_members, const uint32_t* crc, uint8_t* slots,
                         uint32_t slot_stride, uint32_t* sizes, uint32_t* tokens, hipStream_t st);
void launch_compact_members(const uint8_t* slots, uint32_t slot_stride, const uint32_t* sizes, const uint64_t* off, uint32_t n_members,
                            uint8_t* out, hipStream_t st);

// ---- tags ---------------------------------------------------------------------------------------
// loc[t*n + i] = byte offset (from record start) of the aux VALUE of requested tag t in row i,
// typ[t*n + i] = its BAM type char (0 = absent).  tags[t] = two tag bytes little-endian.
void launch_tag_locate(const uint8_t* u, const uint64_t* rows, uint64_t n, const uint16_t* tags_dev, int32_t n_tags,
                       uint32_t* loc, uint8_t* typ, uint32_t* err, hipStream_t st);
enum TagColKind : int32_t { TAG_INT32 = 0, TAG_UINT32 = 1, TAG_FLOAT32 = 2, TAG_UTF8 = 3, TAG_LIST = 4 };
// fixed-width tag column (Int32/UInt32/Float32): values + validity words (row0 = global row index of rows[0])
void launch_tag_fixed(const uint8_t* u, const uint64_t* rows, uint64_t row0, uint64_t n, const uint32_t* loc, const uint8_t* typ,
                      int32_t kind, uint32_t* values, uint64_t* valid, uint32_t* err, hipStream_t st);
// utf8 tag column: pass 1 lengths (+validity), pass 2 scatter
void launch_tag_utf8_len(const uint8_t* u, const uint64_t* rows, uint64_t row0, uint64_t n, const uint32_t* loc, const uint8_t* typ,
                         uint32_t* len, uint64_t* valid, uint32_t* err, hipStream_t st);
void launch_tag_utf8_scatter(const uint8_t* u, const uint64_t* rows, uint64_t n, const uint32_t* loc, const uint8_t* typ,
                             const uint64_t* off64, const uint64_t* valid, uint64_t row0, uint8_t* dst, hipStream_t st);
// list tag column: elem: 0 i8,1 u8,2 i16,3 u16,4 i32,5 u32,6 f32
void launch_tag_list_len(const uint8_t* u, const uint64_t* rows, uint64_t row0, uint64_t n, const uint32_t* loc, const uint8_t* typ,
                         int32_t elem, uint32_t* len, uint64_t* valid, uint32_t* err, hipStream_t st);
void launch_tag_list_scatter(const uint8_t* u, const uint64_t* rows, uint64_t n, const uint32_t* loc, const uint8_t* typ,
                             int32_t elem, const uint64_t* off64, uint8_t* dst, uint32_t* err, hipStream_t st);

// ---- FASTQ (fastq_kernels.hip) -------------------------------------------------------------------
void launch_fastq_sync(const uint8_t* u, uint64_t start, uint64_t ulen, const uint64_t* win_end, const uint64_t* win_coff,
                       const uint64_t* win_next, uint32_t n_win, uint64_t end_comp, int check_end, unsigned long long* result,
                       hipStream_t st);
uint64_t nl_chunks(uint64_t lo, uint64_t hi);
void launch_nl_count(const uint8_t* u, uint64_t lo, uint64_t hi, uint32_t* cnt, hipStream_t st, bool add = false);
void launch_nl_lower_bound(const uint64_t* nl, uint64_t n, uint64_t x, unsigned long long* out, hipStream_t st);
void launch_nl_write(const uint8_t* u, uint64_t lo, uint64_t hi, const uint64_t* base, uint64_t* nl, hipStream_t st);
void launch_fastq_count_owned(const uint64_t* nl, uint64_t n_nl, uint64_t x0, uint64_t eof, uint64_t limit_off,
                              unsigned long long* result, hipStream_t st);
// FASTQ records -> the four Utf8 columns in two passes (the scheme of bam_rows.hip: no per-row length / source arrays):
//   pass 1  one record per lane: field lengths from the newline index, description validity, '@' / '+' checks, the SUM of every
//           projected column's lengths per 256-row tile; launch_tile_scan turns the sums into tile bases + column totals
//   pass 2  one workgroup per tile: the geometry is recomputed, scanned inside the tile, the per-batch int32 offsets are written
//           from that scan and the bytes are copied by 16-lane groups (16-byte chunks, two rows in flight per group)
// k: 0 name, 1 description, 2 sequence, 3 quality.
struct FqCols {
  uint8_t* val[4]; int32_t* off32[4]; uint64_t* base[4];
  uint64_t* v_desc;
  uint32_t want;
};
constexpr int TS_GROUP = 1024;   // tile sums are scanned in groups of 1024 tiles (bam_rows.hip)
// tile_sums: 6 x (n_tiles + 1) entries followed by 6 x (n_groups + 1) group totals (bam_rows_scratch_elems sizes it)
void launch_tile_scan(uint64_t* tile_sums, uint64_t n_tiles, uint32_t want, hipStream_t st);
void launch_fastq_pass1(const uint8_t* u, uint64_t x0, uint64_t eof, const uint64_t* nl, uint64_t n_nl, uint64_t n_rec, FqCols c,
                        uint64_t* tile_sums, uint32_t* err, hipStream_t st);
void launch_fastq_pass2(const uint8_t* u, uint64_t x0, uint64_t eof, const uint64_t* nl, uint64_t n_nl, uint64_t n_rec, FqCols c,
                        uint32_t batch_size, uint32_t phase, const uint64_t* tile_sums, hipStream_t st);
// total_bytes = off64[n] (the caller has just read it): the average row length picks the kernel shape
void launch_scatter_ranges(const uint8_t* u, const uint64_t* src, uint64_t n, const uint64_t* off64, uint8_t* dst, uint64_t total_bytes,
                           hipStream_t st);

}  // namespace bioscan
