//! DataFusion scan operators for BGZF BAM / VCF / FASTQ on AMD MI355X, as a drop-in for the providers of
//! `datafusion-bio-format-{bam,vcf,fastq}`.
//!
//! Same constructors, same `TableProvider` / `ExecutionPlan` behaviour (schema, `supports_filters_pushdown`, partition
//! planning, batches of exactly `batch_size` rows); `execute(partition)` hands the partition to `libbioscan.so`, which
//! inflates, frames and extracts it on the GPU and returns Arrow batches through the Arrow C Data Interface.
//!
//! ```no_run
//! use datafusion::prelude::*;
//! use datafusion_bio_format_gpu::{BamTableProvider, register_vcf_udfs};
//! # async fn demo() -> datafusion::common::Result<()> {
//! let ctx = SessionContext::new();
//! let bam = BamTableProvider::new("reads.bam".into(), None, true, None, false, true, 100, None).await?;
//! ctx.register_table("reads", std::sync::Arc::new(bam))?;
//! register_vcf_udfs(&ctx);
//! ctx.sql("SELECT chrom, COUNT(*) FROM reads WHERE mapping_quality >= 30 GROUP BY chrom").await?.show().await?;
//! # Ok(()) }
//! ```
pub mod ffi;

mod bam;
mod exec;
mod fastq;
mod filters;
mod handles;
mod udfs;
mod vcf;
mod write;

pub use bam::{BamOptions, BamTableProvider};
pub use exec::BioscanExec;
pub use fastq::FastqTableProvider;
pub use udfs::{list_and_udf, list_avg_udf, list_gte_udf, list_lte_udf, register_vcf_udfs, vcf_ac_udf, vcf_af_udf, vcf_an_udf, vcf_set_gts_udf};
pub use vcf::VcfTableProvider;
pub use write::BamWriteExec;

/// Object-storage options of the reference constructors.  Accepted for signature compatibility; the GPU path reads
/// local files only (remote objects are outside the scan path this crate replaces).
#[derive(Debug, Clone, Default)]
pub struct ObjectStorageOptions;

/// `true` for an input the GPU path does not take: a gzip file that is not BGZF (one serial DEFLATE stream -- no blocks to
/// hand to waves; `bioscan_fastq_open` / `bioscan_vcf_open` refuse it).  The reference reads such files with its sequential
/// fallback reader (bio-format-fastq/src/physical_exec.rs:94-138, bio-format-vcf/src/storage.rs:117-122; tests
/// `multimember_gz_test.rs`), so a session that may meet them keeps the reference provider for exactly those paths:
///
/// ```ignore
/// let table: Arc<dyn TableProvider> = if datafusion_bio_format_gpu::needs_reference_reader(&path)? {
///     Arc::new(datafusion_bio_format_fastq::table_provider::FastqTableProvider::new(path, None)?)
/// } else {
///     Arc::new(datafusion_bio_format_gpu::FastqTableProvider::new(path, None)?)
/// };
/// ```
pub fn needs_reference_reader(path: &str) -> std::io::Result<bool> {
    use std::io::Read;
    let mut head = [0u8; 18];
    let n = std::fs::File::open(path)?.read(&mut head)?;
    if n < 4 || head[0] != 0x1f || head[1] != 0x8b {
        return Ok(false); // not gzip at all: plain text, handled by the GPU path's byte-range partitions
    }
    // BGZF = gzip with FEXTRA and a "BC" subfield of length 2 first in the extra field (SAM spec 4.1)
    let bgzf = n >= 18 && head[3] & 4 != 0 && head[12] == b'B' && head[13] == b'C' && head[14] == 2 && head[15] == 0;
    Ok(!bgzf)
}
