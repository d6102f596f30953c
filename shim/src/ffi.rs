//! Raw bindings of `include/bioscan.h` -- one declaration per entry point of the header, same order.
//! `tests/test_cpu_shim_abi.py` fails when a function of the header is missing here (or the reverse).
#![allow(non_camel_case_types)]

use arrow::ffi::{FFI_ArrowArray, FFI_ArrowSchema};
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct bioscan_provider {
    _private: [u8; 0],
}
#[repr(C)]
pub struct bioscan_plan {
    _private: [u8; 0],
}
#[repr(C)]
pub struct bioscan_stream {
    _private: [u8; 0],
}

#[repr(C)]
pub struct bioscan_bam_writer {
    _private: [u8; 0],
}

#[repr(C)]
pub struct bioscan_bam_options {
    pub coordinate_system_zero_based: i32,
    pub tag_fields: *const *const c_char,
    pub n_tag_fields: i32,
    pub binary_cigar: i32,
    pub infer_tag_types: i32,
    pub infer_tag_sample_size: i32,
    pub tag_type_hints: *const *const c_char,
    pub n_tag_type_hints: i32,
    pub index_path: *const c_char,
    pub device_id: i32,
    pub chunk_members: i32,
}

#[repr(C)]
pub struct bioscan_vcf_options {
    pub device_id: i32,
    pub coordinate_system_zero_based: i32,
    pub has_info_fields: i32,
    pub info_fields: *const *const c_char,
    pub n_info_fields: i32,
    pub has_format_fields: i32,
    pub format_fields: *const *const c_char,
    pub n_format_fields: i32,
    pub has_samples: i32,
    pub samples: *const *const c_char,
    pub n_samples: i32,
    pub index_path: *const c_char,
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct bioscan_udf_stats {
    pub n_rows: u64,
    pub n_elements: u64,
    pub count_a: u64,
    pub count_b: u64,
    pub sum: f64,
    pub ms_kernel: f64,
}

pub const BIOSCAN_OP_EQ: i32 = 0;
pub const BIOSCAN_OP_NE: i32 = 1;
pub const BIOSCAN_OP_LT: i32 = 2;
pub const BIOSCAN_OP_LE: i32 = 3;
pub const BIOSCAN_OP_GT: i32 = 4;
pub const BIOSCAN_OP_GE: i32 = 5;
pub const BIOSCAN_OP_BETWEEN: i32 = 6;
pub const BIOSCAN_OP_NOT_BETWEEN: i32 = 7;
pub const BIOSCAN_OP_IN: i32 = 8;
pub const BIOSCAN_OP_NOT_IN: i32 = 9;

pub const BIOSCAN_LIT_NULL: i32 = 0;
pub const BIOSCAN_LIT_INT: i32 = 1;
pub const BIOSCAN_LIT_FLOAT: i32 = 2;
pub const BIOSCAN_LIT_STR: i32 = 3;

#[repr(C)]
pub struct bioscan_literal {
    pub kind: i32,
    pub i: i64,
    pub f: f64,
    pub s: *const c_char,
}

#[repr(C)]
pub struct bioscan_filter {
    pub column: *const c_char,
    pub op: i32,
    pub values: *const bioscan_literal,
    pub n_values: i32,
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct bioscan_scan_stats {
    pub n_blocks: u64,
    pub compressed_bytes: u64,
    pub inflated_bytes: u64,
    pub arrow_bytes: u64,
    pub n_records: u64,
    pub n_rows: u64,
    pub ms_h2d: f64,
    pub ms_frame: f64,
    pub ms_inflate: f64,
    pub ms_chain: f64,
    pub ms_extract: f64,
    pub ms_total_gpu: f64,
    pub ms_crc: f64,
    pub ms_keys: f64,
    pub ms_select: f64,
    pub ms_wall: f64,
    pub chain_iterations: u64,
}

unsafe extern "C" {
    pub fn bioscan_bam_options_default(o: *mut bioscan_bam_options);
    pub fn bioscan_bam_open(path: *const c_char, opts: *const bioscan_bam_options, out: *mut *mut bioscan_provider) -> c_int;
    pub fn bioscan_fastq_open(path: *const c_char, device_id: i32, out: *mut *mut bioscan_provider) -> c_int;
    pub fn bioscan_vcf_options_default(o: *mut bioscan_vcf_options);
    pub fn bioscan_vcf_open(path: *const c_char, opts: *const bioscan_vcf_options, out: *mut *mut bioscan_provider) -> c_int;

    pub fn bioscan_udf_list_avg(input: *const FFI_ArrowArray, in_schema: *const FFI_ArrowSchema, device_id: i32,
                                out: *mut FFI_ArrowArray, out_schema: *mut FFI_ArrowSchema) -> c_int;
    pub fn bioscan_udf_list_cmp(input: *const FFI_ArrowArray, in_schema: *const FFI_ArrowSchema, op: i32, threshold: f64, device_id: i32,
                                out: *mut FFI_ArrowArray, out_schema: *mut FFI_ArrowSchema) -> c_int;
    pub fn bioscan_udf_list_and(a: *const FFI_ArrowArray, a_schema: *const FFI_ArrowSchema, b: *const FFI_ArrowArray,
                                b_schema: *const FFI_ArrowSchema, device_id: i32, out: *mut FFI_ArrowArray,
                                out_schema: *mut FFI_ArrowSchema) -> c_int;
    pub fn bioscan_udf_vcf_set_gts(gt: *const FFI_ArrowArray, gt_schema: *const FFI_ArrowSchema, mask: *const FFI_ArrowArray,
                                   mask_schema: *const FFI_ArrowSchema, replacement: *const c_char, device_id: i32,
                                   out: *mut FFI_ArrowArray, out_schema: *mut FFI_ArrowSchema) -> c_int;
    pub fn bioscan_udf_vcf_allele_stats(gt: *const FFI_ArrowArray, gt_schema: *const FFI_ArrowSchema, alt: *const FFI_ArrowArray,
                                        alt_schema: *const FFI_ArrowSchema, which: i32, device_id: i32, out: *mut FFI_ArrowArray,
                                        out_schema: *mut FFI_ArrowSchema) -> c_int;
    pub fn bioscan_stream_list_udf(s: *mut bioscan_stream, field: *const c_char, udf: i32, threshold: f64,
                                   out: *mut bioscan_udf_stats) -> c_int;

    pub fn bioscan_schema(p: *const bioscan_provider, out: *mut FFI_ArrowSchema) -> c_int;
    pub fn bioscan_supports_filters_pushdown(p: *const bioscan_provider, filters: *const bioscan_filter, n_filters: i32,
                                             out: *mut i32) -> c_int;
    pub fn bioscan_scan(p: *const bioscan_provider, projection: *const i32, n_projection: i32, filters: *const bioscan_filter,
                        n_filters: i32, limit: i64, target_partitions: i32, out: *mut *mut bioscan_plan) -> c_int;
    pub fn bioscan_plan_num_partitions(plan: *const bioscan_plan) -> i32;
    pub fn bioscan_plan_schema(plan: *const bioscan_plan, out: *mut FFI_ArrowSchema) -> c_int;
    pub fn bioscan_plan_display(plan: *const bioscan_plan, buf: *mut c_char, cap: i32) -> i32;
    pub fn bioscan_plan_partition_desc(plan: *const bioscan_plan, partition: i32, buf: *mut c_char, cap: i32) -> i32;
    pub fn bioscan_execute(plan: *const bioscan_plan, partition: i32, batch_size: i32, out: *mut *mut bioscan_stream) -> c_int;
    pub fn bioscan_next(s: *mut bioscan_stream, out: *mut FFI_ArrowArray, has_batch: *mut i32) -> c_int;
    pub fn bioscan_stream_close(s: *mut bioscan_stream);
    pub fn bioscan_plan_close(plan: *mut bioscan_plan);
    pub fn bioscan_provider_close(p: *mut bioscan_provider);
    pub fn bioscan_last_error() -> *const c_char;

    pub fn bioscan_provider_make_resident(p: *mut bioscan_provider) -> c_int;
    pub fn bioscan_provider_set_chunk_members(p: *mut bioscan_provider, chunk_members: i32) -> c_int;
    pub fn bioscan_scan_devices(p: *const bioscan_provider, projection: *const i32, n_projection: i32,
                                filters: *const bioscan_filter, n_filters: i32, limit: i64, target_partitions: i32,
                                device_ids: *const i32, n_devices: i32, out: *mut *mut bioscan_plan) -> c_int;
    pub fn bioscan_plan_partition_device(plan: *const bioscan_plan, partition: i32) -> i32;
    pub fn bioscan_plan_make_resident(plan: *const bioscan_plan, partitions: *const i32, n_partitions: i32) -> c_int;
    pub fn bioscan_provider_resident_range(p: *const bioscan_provider, device_id: i32, lo: *mut u64, hi: *mut u64) -> c_int;
    pub fn bioscan_execute_device(plan: *const bioscan_plan, partition: i32, batch_size: i32, stats: *mut bioscan_scan_stats,
                                  out: *mut *mut bioscan_stream) -> c_int;

    pub fn bioscan_bam_writer_open(path: *const c_char, header_text: *const c_char, ref_names: *const *const c_char,
                                   ref_lengths: *const i64, n_ref: i32, coordinate_system_zero_based: i32, device_id: i32,
                                   out: *mut *mut bioscan_bam_writer) -> c_int;
    pub fn bioscan_bam_writer_open_schema(path: *const c_char, schema: *const FFI_ArrowSchema, sort_on_write: i32, device_id: i32,
                                          out: *mut *mut bioscan_bam_writer) -> c_int;
    pub fn bioscan_bam_header_from_schema(schema: *const FFI_ArrowSchema, sort_on_write: i32, header_text: *mut *mut c_char) -> c_int;
    pub fn bioscan_bam_writer_write(w: *mut bioscan_bam_writer, batch: *const FFI_ArrowArray, schema: *const FFI_ArrowSchema) -> c_int;
    pub fn bioscan_bam_writer_finish(w: *mut bioscan_bam_writer, n_records: *mut u64, n_members: *mut u64, n_bytes: *mut u64) -> c_int;
    pub fn bioscan_bam_writer_close(w: *mut bioscan_bam_writer);
    pub fn bioscan_bgzf_deflate(data: *const u8, len: usize, device_id: i32, add_eof: i32, out: *mut *mut u8, out_len: *mut usize,
                                kernel_ms: *mut f64) -> c_int;

    pub fn bioscan_bgzf_inflate(data: *const u8, len: usize, device_id: i32, check_crc: i32, out: *mut *mut u8,
                                out_len: *mut usize, kernel_ms: *mut f64) -> c_int;
    pub fn bioscan_free(p: *mut c_void);

    pub fn bioscan_debug_balance_partitions(n: i32, chroms: *const *const c_char, region_start: *const u64,
                                            region_end: *const u64, est_bytes: *const u64, contig_len: *const u64,
                                            unmapped: *const u64, bins: *const *const u64, n_bins: *const i32, leaf_span: u64,
                                            target_partitions: i32, buf: *mut c_char, cap: i32) -> i32;
    pub fn bioscan_debug_plan_full_scan(bai_path: *const c_char, n_ref: i32, ref_names: *const *const c_char,
                                        ref_lengths: *const i64, target_partitions: i32, buf: *mut c_char, cap: i32) -> i32;
    pub fn bioscan_debug_extract_regions(filters: *const bioscan_filter, n_filters: i32, zero_based: i32, buf: *mut c_char, cap: i32) -> i32;
    pub fn bioscan_debug_shard_partitions(weights: *const u64, n: i32, world: i32, run_of: *mut i32) -> i32;
    pub fn bioscan_device_check(device_id: i32, name_buf: *mut c_char, cap: i32) -> c_int;
}
