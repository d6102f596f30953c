//! `BamExec` / `VcfExec` / `FastqExec` of the reference as one `ExecutionPlan` over a `bioscan_plan`
//! (bio-format-bam/src/physical_exec.rs:39-173, bio-format-vcf/src/physical_exec.rs:2539-2690,
//! bio-format-fastq/src/physical_exec.rs:262-330).
use crate::ffi;
use crate::filters::FilterSet;
use crate::handles::{PlanHandle, ProviderHandle, StreamHandle, check, last_error};
use arrow::array::{Array, RecordBatch, RecordBatchOptions, StructArray};
use arrow::datatypes::{Schema, SchemaRef};
use arrow::ffi::{FFI_ArrowArray, FFI_ArrowSchema, from_ffi};
use datafusion::common::DataFusionError;
use datafusion::execution::{SendableRecordBatchStream, TaskContext};
use datafusion::logical_expr::Expr;
use datafusion::physical_expr::EquivalenceProperties;
use datafusion::physical_plan::execution_plan::{Boundedness, EmissionType};
use datafusion::physical_plan::stream::RecordBatchStreamAdapter;
use datafusion::physical_plan::{DisplayAs, DisplayFormatType, ExecutionPlan, Partitioning, PlanProperties};
use std::any::Any;
use std::fmt::{Debug, Formatter};
use std::sync::Arc;

/// Schema of a provider or plan through the Arrow C Data Interface (metadata included).
pub(crate) fn import_schema(fill: impl FnOnce(*mut FFI_ArrowSchema) -> i32) -> datafusion::common::Result<SchemaRef> {
    let mut s = FFI_ArrowSchema::empty();
    check(fill(&mut s))?;
    let schema = Schema::try_from(&s).map_err(|e| DataFusionError::ArrowError(Box::new(e), None))?;
    Ok(Arc::new(schema))
}

pub struct BioscanExec {
    name: &'static str,
    plan: Arc<PlanHandle>,
    schema: SchemaRef,
    cache: Arc<PlanProperties>,
    display: String,
}

impl BioscanExec {
    /// `TableProvider::scan` of all three providers: forwards projection, push-down candidates, limit and
    /// `target_partitions` to `bioscan_scan` (or `bioscan_scan_devices` when the provider was given several GPUs).
    pub(crate) fn plan(
        name: &'static str,
        provider: &Arc<ProviderHandle>,
        projection: Option<&Vec<usize>>,
        filters: &[Expr],
        limit: Option<usize>,
        target_partitions: usize,
        device_ids: &[i32],
        emission: EmissionType,
    ) -> datafusion::common::Result<Arc<dyn ExecutionPlan>> {
        let refs: Vec<&Expr> = filters.iter().collect();
        let set = FilterSet::new(&refs);
        let proj: Option<Vec<i32>> = projection.map(|p| p.iter().map(|&i| i as i32).collect());
        // a non-null pointer with zero entries is the empty projection (COUNT(*) batches)
        let empty: [i32; 1] = [0];
        let (pp, pn) = match &proj {
            Some(v) if v.is_empty() => (empty.as_ptr(), 0),
            Some(v) => (v.as_ptr(), v.len() as i32),
            None => (std::ptr::null(), 0),
        };
        let mut raw: *mut ffi::bioscan_plan = std::ptr::null_mut();
        let lim = limit.map(|l| l as i64).unwrap_or(-1);
        let rc = unsafe {
            if device_ids.len() > 1 {
                ffi::bioscan_scan_devices(provider.0, pp, pn, set.raw.as_ptr(), set.raw.len() as i32, lim, target_partitions as i32,
                                          device_ids.as_ptr(), device_ids.len() as i32, &mut raw)
            } else {
                ffi::bioscan_scan(provider.0, pp, pn, set.raw.as_ptr(), set.raw.len() as i32, lim, target_partitions as i32, &mut raw)
            }
        };
        check(rc)?;
        let plan = Arc::new(PlanHandle { raw, _provider: provider.clone() });
        let schema = import_schema(|s| unsafe { ffi::bioscan_plan_schema(plan.raw, s) })?;
        let n = unsafe { ffi::bioscan_plan_num_partitions(plan.raw) }.max(0) as usize;
        if n == 0 {
            // unsatisfiable genomic bounds / LIMIT 0: EmptyExec, as the reference returns (vcf/src/table_provider.rs:1265-1269)
            return Ok(Arc::new(datafusion::physical_plan::empty::EmptyExec::new(schema)));
        }
        let mut buf = vec![0u8; 4096];
        let len = unsafe { ffi::bioscan_plan_display(plan.raw, buf.as_mut_ptr() as *mut _, buf.len() as i32) }.max(0) as usize;
        let display = String::from_utf8_lossy(&buf[..len.min(buf.len() - 1)]).into_owned();
        let cache = Arc::new(PlanProperties::new(
            EquivalenceProperties::new(schema.clone()),
            Partitioning::UnknownPartitioning(n),
            emission,
            Boundedness::Bounded,
        ));
        Ok(Arc::new(BioscanExec { name, plan, schema, cache, display }))
    }
}

impl Debug for BioscanExec {
    fn fmt(&self, f: &mut Formatter<'_>) -> std::fmt::Result {
        f.write_str(&self.display)
    }
}

impl DisplayAs for BioscanExec {
    fn fmt_as(&self, _t: DisplayFormatType, f: &mut Formatter) -> std::fmt::Result {
        f.write_str(&self.display) // "BamExec: projection=[...]" (bam/src/physical_exec.rs:66-82), built by the library
    }
}

impl ExecutionPlan for BioscanExec {
    fn name(&self) -> &str {
        self.name
    }
    fn as_any(&self) -> &dyn Any {
        self
    }
    fn properties(&self) -> &Arc<PlanProperties> {
        &self.cache
    }
    fn children(&self) -> Vec<&Arc<dyn ExecutionPlan>> {
        vec![]
    }
    fn with_new_children(self: Arc<Self>, _children: Vec<Arc<dyn ExecutionPlan>>) -> datafusion::common::Result<Arc<dyn ExecutionPlan>> {
        Ok(self)
    }

    fn execute(&self, partition: usize, context: Arc<TaskContext>) -> datafusion::common::Result<SendableRecordBatchStream> {
        let batch_size = context.session_config().batch_size();
        let mut raw: *mut ffi::bioscan_stream = std::ptr::null_mut();
        check(unsafe { ffi::bioscan_execute(self.plan.raw, partition as i32, batch_size as i32, &mut raw) })?;
        let stream = StreamHandle { raw, _plan: self.plan.clone() };
        let schema = self.schema.clone();
        // Pull-based, blocking on the polling worker, one OS thread per partition: sync_batch_stream
        // (bio-format-core/src/sync_stream.rs:34-43).  The GPU runs chunk k + 1 while chunk k is copied to the host.
        let state = (stream, schema.clone(), false);
        let s = futures::stream::unfold(state, |(stream, schema, done)| async move {
            if done {
                return None;
            }
            match next_batch(&stream, &schema) {
                Ok(Some(b)) => Some((Ok(b), (stream, schema, false))),
                Ok(None) => None,
                Err(e) => Some((Err(e), (stream, schema, true))), // the error is yielded as an item, then the stream ends
            }
        });
        Ok(Box::pin(RecordBatchStreamAdapter::new(self.schema.clone(), s)))
    }
}

fn next_batch(stream: &StreamHandle, schema: &SchemaRef) -> datafusion::common::Result<Option<RecordBatch>> {
    let mut array = FFI_ArrowArray::empty();
    let mut has: i32 = 0;
    if unsafe { ffi::bioscan_next(stream.raw, &mut array, &mut has) } != 0 {
        return Err(last_error());
    }
    if has == 0 {
        return Ok(None);
    }
    // the exported array is a struct of the projected columns; its buffers stay owned by the library until arrow drops them
    let ffi_schema = FFI_ArrowSchema::try_from(schema.as_ref()).map_err(|e| DataFusionError::ArrowError(Box::new(e), None))?;
    let ffi_struct = FFI_ArrowSchema::try_from(&arrow::datatypes::DataType::Struct(schema.fields().clone()))
        .map_err(|e| DataFusionError::ArrowError(Box::new(e), None))?;
    let _ = ffi_schema;
    let data = unsafe { from_ffi(array, &ffi_struct) }.map_err(|e| DataFusionError::ArrowError(Box::new(e), None))?;
    let rows = data.len();
    let st = StructArray::from(data);
    let columns = st.columns().to_vec();
    // zero-column plans carry only a row count (bio-format-core/src/alignment_utils.rs:360-363)
    RecordBatch::try_new_with_options(schema.clone(), columns, &RecordBatchOptions::new().with_row_count(Some(rows)))
        .map(Some)
        .map_err(|e| DataFusionError::ArrowError(Box::new(e), None))
}

/// `supports_filters_pushdown` of all three providers (bam/src/table_provider.rs:941-962): `Inexact` for what the library
/// accepts, `Unsupported` otherwise -- DataFusion re-applies every predicate above the scan.
pub(crate) fn pushdown(
    provider: &ProviderHandle,
    filters: &[&Expr],
) -> datafusion::common::Result<Vec<datafusion::logical_expr::TableProviderFilterPushDown>> {
    use datafusion::logical_expr::TableProviderFilterPushDown as P;
    let set = FilterSet::new(filters);
    let mut answers = vec![0i32; set.raw.len().max(1)];
    check(unsafe { ffi::bioscan_supports_filters_pushdown(provider.0, set.raw.as_ptr(), set.raw.len() as i32, answers.as_mut_ptr()) })?;
    let mut out = vec![P::Unsupported; filters.len()];
    for (k, &src) in set.source.iter().enumerate() {
        if answers[k] != 0 {
            out[src] = P::Inexact;
        }
    }
    Ok(out)
}
