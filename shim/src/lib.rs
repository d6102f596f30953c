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

pub use bam::{BamOptions, BamTableProvider};
pub use exec::BioscanExec;
pub use fastq::FastqTableProvider;
pub use udfs::{list_and_udf, list_avg_udf, list_gte_udf, list_lte_udf, register_vcf_udfs, vcf_set_gts_udf};
pub use vcf::VcfTableProvider;

/// Object-storage options of the reference constructors.  Accepted for signature compatibility; the GPU path reads
/// local files only (remote objects are outside the scan path this crate replaces).
#[derive(Debug, Clone, Default)]
pub struct ObjectStorageOptions;
