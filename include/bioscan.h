/* bioscan.h -- C ABI of the MI355X-native BGZF -> Arrow scan engine (libbioscan.so).
 *
 * This is the drop-in boundary for ONE path of biodatageeks/datafusion-bio-formats: the
 * partitioned DataFusion TableProvider / ExecutionPlan scan of BGZF files.  Every entry
 * point names the reference interface it replaces (paths relative to
 * /root/reference/datafusion).  Plain pointers and sizes only; results cross the boundary
 * through the Arrow C Data Interface (struct ArrowSchema / struct ArrowArray), which is
 * what `arrow::ffi` (Rust), pyarrow and every other Arrow binding import zero-copy.
 *
 * Threading contract (mirrors bio-format-core/src/sync_stream.rs:7-33): distinct streams
 * may be driven from distinct threads; one stream is used by one thread at a time.
 * All functions return 0 on success; on failure they return non-zero and
 * bioscan_last_error() (thread-local) describes the failure -- the analogue of the
 * reference yielding DataFusionError::Execution(..) as a stream item
 * (bio-format-bam/src/physical_exec.rs:567-571).
 */
#ifndef BIOSCAN_H
#define BIOSCAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct ArrowSchema; /* Arrow C Data Interface (arrow.apache.org/docs/format/CDataInterface.html) */
struct ArrowArray;

typedef struct bioscan_provider bioscan_provider; /* BamTableProvider            */
typedef struct bioscan_plan bioscan_plan;         /* BamExec (an ExecutionPlan)  */
typedef struct bioscan_stream bioscan_stream;     /* SendableRecordBatchStream   */

/* ---- BamTableProvider::new (bio-format-bam/src/table_provider.rs:381-529) ------------
 * Same knobs as the reference constructor's positional arguments; object-storage options
 * are out of scope (local files only).  `index_path` NULL = auto-discover `<path>.bai` /
 * `<stem>.bai` (bio-format-core/src/index_utils.rs:68-83); "" = force no index. */
typedef struct bioscan_bam_options {
  int32_t coordinate_system_zero_based; /* default 1 */
  const char* const* tag_fields;        /* NULL = no tag columns */
  int32_t n_tag_fields;
  int32_t binary_cigar;                 /* 0 = Utf8 cigar */
  int32_t infer_tag_types;              /* sample the file for tags outside the SAM registry */
  int32_t infer_tag_sample_size;        /* reference default 100 */
  const char* const* tag_type_hints;    /* "XY:i", "pa:B:f" ...; NULL = none */
  int32_t n_tag_type_hints;
  const char* index_path;
  int32_t device_id;                    /* HIP device ordinal this provider's scans run on */
  int32_t chunk_members;                /* BGZF members per pipeline chunk of a stream (bounds the HBM and host memory of
                                           one execute); 0 = default (16384, or BIOSCAN_CHUNK_MEMBERS) */
} bioscan_bam_options;

void bioscan_bam_options_default(bioscan_bam_options* o);

int bioscan_bam_open(const char* path, const bioscan_bam_options* opts, bioscan_provider** out);

/* FastqTableProvider::new (bio-format-fastq/src/table_provider.rs:49-66): 4-column schema
 * (name, description, sequence, quality_scores).  The handle is used with the same
 * bioscan_scan / bioscan_execute / bioscan_next entry points; `filters` are ignored by the FASTQ scan
 * (bio-format-fastq/src/table_provider.rs:94 `_filters`) and `limit` stops each partition
 * (physical_exec.rs:416).  Strategy = detect_local_strategy (physical_exec.rs:94-138): BGZF + `.gzi`
 * -> block-range partitions, plain file -> byte ranges, BGZF without index -> one partition. */
int bioscan_fastq_open(const char* path, int32_t device_id, bioscan_provider** out);

/* ---- VcfTableProvider::new_with_samples (bio-format-vcf/src/table_provider.rs:796-814, 849-1097) --------
 * Text VCF, BGZF-compressed (`.vcf.gz` / `.vcf.bgz`) or uncompressed.  Schema = 8 core columns + one column per
 * INFO tag + FORMAT columns (single-sample source: top-level columns; multi-sample source: one
 * `genotypes: Struct<tag: List<T>>` column), determine_schema_from_header (table_provider.rs:91-338).
 * `has_*` = 0 means "None" in the reference (all header INFO / FORMAT tags, all samples); with has_* = 1 the
 * array (possibly empty) is the explicit selection.  `index_path` NULL = auto-discover `<path>.tbi`
 * (bio-format-core/src/index_utils.rs:85-100); "" = no index.  The handle is used with the same
 * bioscan_scan / bioscan_execute / bioscan_next entry points (VcfTableProvider::scan table_provider.rs:1225-1462,
 * VcfExec::execute physical_exec.rs:2612-2690).  `limit` stops each partition after `limit` rows
 * (physical_exec.rs:1114-1116, 2997-2999); limit 0 gives an empty plan (table_provider.rs:1265-1269). */
typedef struct bioscan_vcf_options {
  int32_t device_id;
  int32_t coordinate_system_zero_based; /* default 1 */
  int32_t has_info_fields;
  const char* const* info_fields;
  int32_t n_info_fields;
  int32_t has_format_fields;
  const char* const* format_fields;
  int32_t n_format_fields;
  int32_t has_samples;
  const char* const* samples;
  int32_t n_samples;
  const char* index_path;
} bioscan_vcf_options;
void bioscan_vcf_options_default(bioscan_vcf_options* o);
int bioscan_vcf_open(const char* path, const bioscan_vcf_options* opts, bioscan_provider** out);

/* ---- list UDFs (bio-format-vcf/src/udfs.rs) ----------------------------------------------------------------
 * list_avg  :67-110  List<Int32|Float32> -> Float64 (mean of the non-null elements; NULL list / no element -> NULL)
 * list_gte  :606-650 List<Int32|Float32>, threshold -> List<Boolean> (element NULLs kept, NULL list -> NULL list)
 * list_lte  same loop with <=.
 * Host form: Arrow C Data in (a List array + its schema), Arrow C Data out; the arithmetic runs on `device_id`.
 * op: 0 = list_gte, 1 = list_lte.  The threshold is an Int32 for Int32 lists and is rounded to f32 for Float32 lists. */
int bioscan_udf_list_avg(const struct ArrowArray* in, const struct ArrowSchema* in_schema, int32_t device_id,
                         struct ArrowArray* out, struct ArrowSchema* out_schema);
int bioscan_udf_list_cmp(const struct ArrowArray* in, const struct ArrowSchema* in_schema, int32_t op, double threshold,
                         int32_t device_id, struct ArrowArray* out, struct ArrowSchema* out_schema);
/* list_and :765-850 List<Boolean>, List<Boolean> -> List<Boolean> (SQL three-valued AND over min(len) elements; a NULL list
 * on either side gives a NULL list).  vcf_set_gts :857-953 List<Utf8> genotypes, List<Boolean> mask, replacement ->
 * List<Utf8>: a genotype is replaced where its mask element is false and kept where it is true, NULL or absent. */
int bioscan_udf_list_and(const struct ArrowArray* a, const struct ArrowSchema* a_schema, const struct ArrowArray* b,
                         const struct ArrowSchema* b_schema, int32_t device_id, struct ArrowArray* out, struct ArrowSchema* out_schema);
int bioscan_udf_vcf_set_gts(const struct ArrowArray* gt, const struct ArrowSchema* gt_schema, const struct ArrowArray* mask,
                            const struct ArrowSchema* mask_schema, const char* replacement, int32_t device_id,
                            struct ArrowArray* out, struct ArrowSchema* out_schema);
/* vcf_an / vcf_ac / vcf_af (bio-format-vcf/src/udfs.rs:113-142 parse_gt_alleles, 161-552): allele statistics of a List<Utf8> GT
 * column.  which: 0 = AN -> Int32 (called alleles of the row's non-NULL genotypes; ".", "./." and ".|." are entirely missing,
 * a piece that is "." or not a usize is a missing allele), 1 = AC -> List<Int32> (count of allele 1, 2, ...; the list is as long
 * as the larger of the largest called allele index and -- when `alt`, the provider's pipe-separated Utf8 ALT column, is given
 * and not NULL in the row -- the number of ALT alleles), 2 = AF -> List<Float64> (AC / AN; NULL elements when AN is 0).  A NULL
 * list gives NULL.  `alt` / `alt_schema` may be NULL (the one-argument forms). */
int bioscan_udf_vcf_allele_stats(const struct ArrowArray* gt, const struct ArrowSchema* gt_schema, const struct ArrowArray* alt,
                                 const struct ArrowSchema* alt_schema, int32_t which, int32_t device_id, struct ArrowArray* out,
                                 struct ArrowSchema* out_schema);
/* Device-resident form for a stream returned by bioscan_execute_device: applies the UDF to `genotypes.<field>` (or a
 * top-level List column named `field`) of the whole partition without leaving HBM.  udf: 0 list_avg, 1 list_gte,
 * 2 list_lte.  Reports the kernel time and two checksums the caller can compare with an oracle: for list_avg the
 * number of non-NULL results and their f64 sum (added in row order on the host); for the comparisons the number of
 * true and of NULL elements. */
typedef struct bioscan_udf_stats {
  uint64_t n_rows;
  uint64_t n_elements;
  uint64_t count_a;   /* list_avg: non-NULL results;  cmp: true elements  */
  uint64_t count_b;   /* list_avg: NULL results;      cmp: NULL elements  */
  double sum;         /* list_avg: sum of the results */
  double ms_kernel;
} bioscan_udf_stats;
int bioscan_stream_list_udf(bioscan_stream* s, const char* field, int32_t udf, double threshold, bioscan_udf_stats* out);

/* TableProvider::schema (table_provider.rs:933-935); caller releases the ArrowSchema. */
int bioscan_schema(const bioscan_provider* p, struct ArrowSchema* out);

/* ---- filters handed to TableProvider::scan ---------------------------------------------
 * One entry per conjunct of the WHERE clause, restricted to the shapes the reference can
 * push down (bio-format-core/src/genomic_filter.rs:143-331, record_filter.rs:40-356):
 * column <op> literal, column [NOT] BETWEEN lo AND hi, column [NOT] IN (...). */
enum bioscan_filter_op {
  BIOSCAN_OP_EQ = 0, BIOSCAN_OP_NE, BIOSCAN_OP_LT, BIOSCAN_OP_LE, BIOSCAN_OP_GT, BIOSCAN_OP_GE,
  BIOSCAN_OP_BETWEEN, BIOSCAN_OP_NOT_BETWEEN, BIOSCAN_OP_IN, BIOSCAN_OP_NOT_IN
};
enum bioscan_literal_kind { BIOSCAN_LIT_NULL = 0, BIOSCAN_LIT_INT, BIOSCAN_LIT_FLOAT, BIOSCAN_LIT_STR };
typedef struct bioscan_literal {
  int32_t kind;
  int64_t i;
  double f;
  const char* s;
} bioscan_literal;
typedef struct bioscan_filter {
  const char* column;
  int32_t op;
  const bioscan_literal* values; /* 1 (comparison), 2 (between) or n (in-list) literals */
  int32_t n_values;
} bioscan_filter;

/* TableProvider::supports_filters_pushdown (table_provider.rs:941-962):
 * out[i] = 0 Unsupported, 1 Inexact. */
int bioscan_supports_filters_pushdown(const bioscan_provider* p, const bioscan_filter* filters,
                                      int32_t n_filters, int32_t* out);

/* ---- TableProvider::scan (table_provider.rs:964-1115) -----------------------------------
 * projection NULL = all columns; n_projection 0 with non-NULL pointer = empty projection
 * (COUNT(*) batches).  `limit` < 0 = none (stored, never applied, like BamExec.limit).
 * `target_partitions` = SessionConfig::target_partitions.  The plan may have zero
 * partitions (EmptyExec: unsatisfiable genomic bounds). */
int bioscan_scan(const bioscan_provider* p, const int32_t* projection, int32_t n_projection,
                 const bioscan_filter* filters, int32_t n_filters, int64_t limit,
                 int32_t target_partitions, bioscan_plan** out);

/* ExecutionPlan::properties().partitioning (UnknownPartitioning(n)) */
int32_t bioscan_plan_num_partitions(const bioscan_plan* plan);
/* projected schema of the plan (incl. metadata) */
int bioscan_plan_schema(const bioscan_plan* plan, struct ArrowSchema* out);
/* DisplayAs: "BamExec: projection=[...]" (physical_exec.rs:66-82); returns bytes written */
int32_t bioscan_plan_display(const bioscan_plan* plan, char* buf, int32_t cap);
/* Human-readable partition assignment (regions of PartitionAssignment, for tests/diagnostics):
 * "chrom:start-end[;...]" with '*' markers for unmapped tails. */
int32_t bioscan_plan_partition_desc(const bioscan_plan* plan, int32_t partition, char* buf, int32_t cap);

/* ---- ExecutionPlan::execute(partition, ctx) (physical_exec.rs:108-172) -----------------
 * batch_size = ctx.session_config().batch_size() (reference default 8192). */
int bioscan_execute(const bioscan_plan* plan, int32_t partition, int32_t batch_size, bioscan_stream** out);

/* Stream::poll_next: *has_batch = 0 at end of stream.  The exported array is a struct
 * array of the projected columns with `length` rows; its release callback frees the host
 * staging memory.  Zero-column plans export a zero-child struct whose length is the row
 * count (bio-format-core/src/alignment_utils.rs:360-363). */
int bioscan_next(bioscan_stream* s, struct ArrowArray* out, int32_t* has_batch);

void bioscan_stream_close(bioscan_stream* s);
void bioscan_plan_close(bioscan_plan* plan);
void bioscan_provider_close(bioscan_provider* p);

const char* bioscan_last_error(void);

/* ---- Device-resident entry points (measurement / embedding in a GPU pipeline) -----------
 * Runs the whole partition on the GPU and leaves the Arrow column buffers in HBM, without
 * the D2H export that bioscan_next performs.  `stats` (may be NULL) receives per-stage GPU
 * times measured with HIP events on the stream the kernels ran on. */
typedef struct bioscan_scan_stats {
  uint64_t n_blocks;          /* BGZF blocks inflated */
  uint64_t compressed_bytes;  /* C: bytes of those blocks */
  uint64_t inflated_bytes;    /* U */
  uint64_t arrow_bytes;       /* A: bytes of Arrow buffers produced (values+offsets+validity) */
  uint64_t n_records;         /* records walked */
  uint64_t n_rows;            /* rows emitted (after region / residual filters) */
  double ms_h2d;              /* host -> device copy of the compressed bytes (0 if resident) */
  double ms_frame;            /* BGZF framing */
  double ms_inflate;          /* K1 bgzf_inflate */
  double ms_chain;            /* record boundary scan */
  double ms_extract;          /* field extract + Arrow scatter */
  double ms_total_gpu;        /* sum of the stage times above that ran for this call */
  double ms_crc;              /* K2 bgzf_crc32 (validation, as noodles-bgzf verifies every block) */
  double ms_keys;             /* record key table (refid,pos,end,flag,mapq) */
  double ms_select;           /* row selection (region / tail / residual filters) */
  double ms_wall;             /* host wall-clock of the whole call (allocation + launches + syncs) */
  uint64_t chain_iterations;  /* verify/fix rounds of the record boundary scan (1 = all guesses right) */
} bioscan_scan_stats;

/* Make the file's compressed bytes resident in HBM (idempotent); later executes reuse them. */
int bioscan_provider_make_resident(bioscan_provider* p);

/* BGZF members per pipeline chunk of the streams this provider's plans execute from now on (0 = the default: 16 384, or
 * BIOSCAN_CHUNK_MEMBERS).  A stream inflates, frames and extracts one chunk at a time and carries the record / line cut by a
 * chunk's end to the next: its HBM and host footprint is O(chunk), as the reference's record-at-a-time readers are O(batch)
 * (bio-format-fastq/src/physical_exec.rs:393-465, bio-format-vcf/src/physical_exec.rs:912-1198).  bioscan_bam_options has the
 * same knob at open; this one also reaches FASTQ and VCF providers, whose constructors take no such option. */
int bioscan_provider_set_chunk_members(bioscan_provider* p, int32_t chunk_members);

/* ---- several GPUs of one node behind ONE plan (DataFusion calls execute(partition) for every partition
 * of one ExecutionPlan object).  Like bioscan_scan, and the plan's partitions are dealt to `device_ids` in contiguous runs
 * in plan order, balanced by PartitionAssignment.total_estimated_bytes -- the rule of partition_byte_ranges_in_order
 * (bio-format-core/src/range_planning.rs:147-195) -- so the devices' outputs concatenated in order reproduce the
 * single-device partition order.  No collective: partitions never exchange data.  Each device receives only the
 * compressed byte range its partitions inflate (uploaded on first use, or ahead of time by bioscan_plan_make_resident).
 * BAM providers; a VCF / FASTQ provider accepts exactly one device. */
int bioscan_scan_devices(const bioscan_provider* p, const int32_t* projection, int32_t n_projection,
                         const bioscan_filter* filters, int32_t n_filters, int64_t limit, int32_t target_partitions,
                         const int32_t* device_ids, int32_t n_devices, bioscan_plan** out);
/* HIP device that executes `partition` (-1: no such partition). */
int32_t bioscan_plan_partition_device(const bioscan_plan* plan, int32_t partition);
/* Upload now what the given partitions (NULL = every partition of the plan) will inflate, each to its own device. */
int bioscan_plan_make_resident(const bioscan_plan* plan, const int32_t* partitions, int32_t n_partitions);
/* Compressed byte range [*lo, *hi) of the file that is resident on `device_id` (both 0: nothing). */
int bioscan_provider_resident_range(const bioscan_provider* p, int32_t device_id, uint64_t* lo, uint64_t* hi);
/* Execute a partition entirely on the device; columns stay in HBM until the stream is closed
 * or drained with bioscan_next. */
int bioscan_execute_device(const bioscan_plan* plan, int32_t partition, int32_t batch_size,
                           bioscan_scan_stats* stats, bioscan_stream** out);

/* ---- write path: BamLocalWriter (bio-format-bam/src/writer.rs:57-283) + batch_to_alignment_records
 * (bio-format-core/src/sam_record_serializer.rs:15-258) + noodles-bgzf Writer, on the GPU ---------------------------
 * open = new + write_header: `header_text` is the SAM header text, ref_names / ref_lengths the @SQ dictionary in order.
 * write = write_records of one RecordBatch (a struct array holding at least name, chrom, start, flags, cigar [Utf8 or
 * Binary], mapping_quality, mate_chrom, mate_start, sequence, quality_scores, template_length -- the reader's own
 * schema): records are serialised, CRC32-summed and DEFLATE-compressed on the device into BGZF members of at most 65280
 * payload bytes (dynamic-Huffman blocks; fixed-code or stored blocks where those are smaller).  finish = the last short member, the BGZF
 * EOF marker, close.  Errors of the reference are kept ("does not fit into 16-bit SAM flags", CIGAR parse errors).
 * Tag columns: every field that carries "bio.bam.tag.tag" metadata becomes an aux field of the record, in schema order, NULL
 * values skipped (build_tag_data, bio-format-core/src/sam_tag_io.rs:109-147); the SAM type comes from "bio.bam.tag.type" ("Z"
 * when absent; "B:<subtype>" for arrays), accepted Arrow storages and every conversion error are arrow_to_sam_tag_value's
 * (:206-656): Int8..Int64 / UInt8..UInt64 for c s i C S I and A, Float32 / Float64 for f, Utf8 for Z H A, List of those for B. */
typedef struct bioscan_bam_writer bioscan_bam_writer;
int bioscan_bam_writer_open(const char* path, const char* header_text, const char* const* ref_names, const int64_t* ref_lengths,
                            int32_t n_ref, int32_t coordinate_system_zero_based, int32_t device_id, bioscan_bam_writer** out);
/* The writer of an INSERT OVERWRITE (TableProvider::insert_into + BamWriteExec, bio-format-bam/src/table_provider.rs:1117-1178,
 * write_exec.rs:72-170, 281-336): everything comes from the Arrow schema of the rows -- the SAM header from its "bio.bam.*"
 * metadata (build_bam_header, bio-format-bam/src/header_builder.rs:42-195: @HD VN default 1.6, SO / GO / SS, @SQ / @RG / @PG
 * from their JSON arrays, @CO), the coordinate system from "bio.coordinate_system_zero_based" (0-based when absent).
 * sort_on_write: 1 -> @HD SO:coordinate, 0 -> SO:unsorted (the provider's override of the schema's value; the sort itself is
 * DataFusion's SortExec above the write plan), -1 -> the schema's own value.  A path ending in ".sam" selects the plain SAM
 * writer in the reference (BamCompressionType::from_path, writer.rs:27-43) and is refused here.
 * bioscan_bam_header_from_schema returns the header text alone (malloc'd, bioscan_free). */
int bioscan_bam_writer_open_schema(const char* path, const struct ArrowSchema* schema, int32_t sort_on_write, int32_t device_id,
                                   bioscan_bam_writer** out);
int bioscan_bam_header_from_schema(const struct ArrowSchema* schema, int32_t sort_on_write, char** header_text);
int bioscan_bam_writer_write(bioscan_bam_writer* w, const struct ArrowArray* batch, const struct ArrowSchema* schema);
int bioscan_bam_writer_finish(bioscan_bam_writer* w, uint64_t* n_records, uint64_t* n_members, uint64_t* n_bytes);
void bioscan_bam_writer_close(bioscan_bam_writer* w);
/* Kernel-level: compresses a host buffer into BGZF members (<= 65280 payload bytes each; + the EOF marker when add_eof)
 * on the GPU; the result is a malloc'd host buffer (bioscan_free).  Replaces noodles-bgzf Writer + libdeflate's compress. */
int bioscan_bgzf_deflate(const uint8_t* data, size_t len, int32_t device_id, int32_t add_eof, uint8_t** out, size_t* out_len,
                         double* kernel_ms);

/* ---- Kernel-level entry point: K1 alone (parity tests against libdeflate/zlib) ----------
 * Inflates every BGZF member of a host buffer on the GPU and returns the concatenated
 * payload in a malloc'd host buffer (caller frees with bioscan_free).
 * Replaces noodles-bgzf Reader::read_block + libdeflate (bio-format-bam/src/storage.rs:161-169). */
int bioscan_bgzf_inflate(const uint8_t* data, size_t len, int32_t device_id, int32_t check_crc,
                         uint8_t** out, size_t* out_len, double* kernel_ms);
void bioscan_free(void* p);

/* ---- Planning test hooks (host only, usable without a GPU) --------------------------------
 * balance_partitions (bio-format-core/src/partition_balancer.rs:61-295) on caller-supplied
 * estimates.  region_start/region_end: 0 = None; contig_len: 0 = None; bins[i] = n_bins[i] sorted
 * non-empty leaf-bin positions.  Output: one line per partition "bytes|chrom:start-end[*];...". */
int32_t bioscan_debug_balance_partitions(int32_t n, const char* const* chroms, const uint64_t* region_start,
                                         const uint64_t* region_end, const uint64_t* est_bytes,
                                         const uint64_t* contig_len, const uint64_t* unmapped,
                                         const uint64_t* const* bins, const int32_t* n_bins, uint64_t leaf_span,
                                         int32_t target_partitions, char* buf, int32_t cap);
/* Full-scan plan of TableProvider::scan (table_provider.rs:1014-1056) from a BAI file and the
 * header's reference names/lengths: estimate_sizes_from_bai + balance_partitions + no-coor
 * partition.  Same output format. */
int32_t bioscan_debug_plan_full_scan(const char* bai_path, int32_t n_ref, const char* const* ref_names,
                                     const int64_t* ref_lengths, int32_t target_partitions, char* buf, int32_t cap);

/* extract_genomic_regions + is_genomic_coordinate_filter (bio-format-core/src/genomic_filter.rs:51-147) on caller-supplied
 * filters: "chrom:start-end;...|unsat=0|genomic=N|residual=M" (regions 1-based inclusive, empty bound = None). */
int32_t bioscan_debug_extract_regions(const bioscan_filter* filters, int32_t n_filters, int32_t zero_based, char* buf, int32_t cap);
/* partition_byte_ranges_in_order (bio-format-core/src/range_planning.rs:147-195) on per-partition byte estimates:
 * run_of[i] = index of the contiguous run (device / rank) partition i belongs to; returns the number of runs. */
int32_t bioscan_debug_shard_partitions(const uint64_t* weights, int32_t n, int32_t world, int32_t* run_of);

/* Library / device probe: returns 0 when a gfx950-capable HIP device is usable. */
int bioscan_device_check(int32_t device_id, char* name_buf, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* BIOSCAN_H */
