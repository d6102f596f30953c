//! `BamWriteExec` (bio-format-bam/src/write_exec.rs:40-336) over `bioscan_bam_writer_*`: the plan `INSERT OVERWRITE` runs.
//!
//! The reference builds the SAM header from the input schema's `bio.bam.*` metadata (`build_bam_header`), converts every
//! batch to noodles records and writes them through noodles' BAM writer.  Here the schema goes to
//! `bioscan_bam_writer_open_schema` (header, `@SQ` dictionary and coordinate system are read from it on the C side, with the
//! provider's sort-order override) and every batch to `bioscan_bam_writer_write`, which serialises, CRC32-sums and
//! DEFLATE-compresses it on the GPU.  `sort_on_write` wraps the input in `SortExec` + `SortPreservingMergeExec` at execution
//! time exactly as the reference does (write_exec.rs:211-246): sorting is DataFusion's, not the writer's.
use crate::ffi;
use crate::handles::{check, cstring};
use arrow::array::{Array, RecordBatch, StructArray, UInt64Array};
use arrow::compute::SortOptions;
use arrow::datatypes::{DataType, Field, Schema, SchemaRef};
use arrow::ffi::{FFI_ArrowSchema, to_ffi};
use datafusion::common::DataFusionError;
use datafusion::execution::{SendableRecordBatchStream, TaskContext};
use datafusion::physical_expr::expressions::Column;
use datafusion::physical_expr::{EquivalenceProperties, LexOrdering, PhysicalSortExpr};
use datafusion::physical_plan::execution_plan::{Boundedness, EmissionType};
use datafusion::physical_plan::sorts::sort::SortExec;
use datafusion::physical_plan::sorts::sort_preserving_merge::SortPreservingMergeExec;
use datafusion::physical_plan::stream::RecordBatchStreamAdapter;
use datafusion::physical_plan::{DisplayAs, DisplayFormatType, Distribution, ExecutionPlan, Partitioning, PlanProperties};
use futures::StreamExt;
use std::any::Any;
use std::fmt::{Debug, Formatter};
use std::sync::Arc;

/// Owning wrapper of a `bioscan_bam_writer`; closed on drop (an unfinished file is left without its EOF marker, like a
/// noodles writer dropped before `finish`).
struct WriterHandle(*mut ffi::bioscan_bam_writer);
unsafe impl Send for WriterHandle {}
impl Drop for WriterHandle {
    fn drop(&mut self) {
        unsafe { ffi::bioscan_bam_writer_close(self.0) }
    }
}

pub struct BamWriteExec {
    input: Arc<dyn ExecutionPlan>,
    output_path: String,
    tag_fields: Vec<String>,
    coordinate_system_zero_based: bool,
    sort_on_write: bool,
    device_id: i32,
    cache: Arc<PlanProperties>,
}

impl BamWriteExec {
    /// Arguments of the reference's `BamWriteExec::new` that still mean something here.  `tag_fields` (the TABLE schema's tag
    /// columns plus the names given to `new_for_write`) and `coordinate_system_zero_based` (the TABLE schema's flag) are what
    /// `insert_into` resolves (table_provider.rs:1131-1154); the C writer reads both from the schema it is handed, so
    /// `writer_layout` below turns them into that schema: the input plan's own metadata does not decide either.
    pub fn new(input: Arc<dyn ExecutionPlan>, output_path: String, tag_fields: Vec<String>, coordinate_system_zero_based: bool, sort_on_write: bool,
               device_id: i32) -> Self {
        let output_schema = Arc::new(Schema::new(vec![Field::new("count", DataType::UInt64, false)]));
        let cache = Arc::new(PlanProperties::new(
            EquivalenceProperties::new(output_schema),
            Partitioning::UnknownPartitioning(1),
            EmissionType::Final,
            Boundedness::Bounded,
        ));
        Self { input, output_path, tag_fields, coordinate_system_zero_based, sort_on_write, device_id, cache }
    }
    pub fn output_path(&self) -> &str {
        &self.output_path
    }
}

impl Debug for BamWriteExec {
    fn fmt(&self, f: &mut Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("BamWriteExec").field("output_path", &self.output_path).field("compression", &"Bgzf").field("tag_fields", &self.tag_fields).finish()
    }
}

impl DisplayAs for BamWriteExec {
    fn fmt_as(&self, _t: DisplayFormatType, f: &mut Formatter) -> std::fmt::Result {
        write!(f, "BamWriteExec: path={}, compression=Bgzf, tag_fields={}", self.output_path, self.tag_fields.len())
    }
}

impl ExecutionPlan for BamWriteExec {
    fn name(&self) -> &str {
        "BamWriteExec"
    }
    fn as_any(&self) -> &dyn Any {
        self
    }
    fn properties(&self) -> &Arc<PlanProperties> {
        &self.cache
    }
    fn required_input_distribution(&self) -> Vec<Distribution> {
        // write_exec.rs:160-168
        if self.sort_on_write { vec![Distribution::UnspecifiedDistribution] } else { vec![Distribution::SinglePartition] }
    }
    fn children(&self) -> Vec<&Arc<dyn ExecutionPlan>> {
        vec![&self.input]
    }
    fn with_new_children(self: Arc<Self>, children: Vec<Arc<dyn ExecutionPlan>>) -> datafusion::common::Result<Arc<dyn ExecutionPlan>> {
        if children.len() != 1 {
            return Err(DataFusionError::Internal("BamWriteExec requires exactly one child".to_string()));
        }
        Ok(Arc::new(BamWriteExec::new(children[0].clone(), self.output_path.clone(), self.tag_fields.clone(), self.coordinate_system_zero_based,
                                      self.sort_on_write, self.device_id)))
    }

    fn execute(&self, _partition: usize, context: Arc<TaskContext>) -> datafusion::common::Result<SendableRecordBatchStream> {
        let input_schema = self.input.schema();
        // write_exec.rs:211-246: chrom, start ascending, nulls last; sort per partition, then merge
        let input = if self.sort_on_write {
            let chrom_idx = input_schema.index_of("chrom").map_err(|_| DataFusionError::Plan("Column 'chrom' not found for sort_on_write".to_string()))?;
            let start_idx = input_schema.index_of("start").map_err(|_| DataFusionError::Plan("Column 'start' not found for sort_on_write".to_string()))?;
            let opts = SortOptions { descending: false, nulls_first: false };
            let sort_exprs = LexOrdering::new(vec![
                PhysicalSortExpr::new(Arc::new(Column::new("chrom", chrom_idx)), opts),
                PhysicalSortExpr::new(Arc::new(Column::new("start", start_idx)), opts),
            ])
            .expect("sort expressions should not be empty");
            let sort_exec = Arc::new(SortExec::new(sort_exprs.clone(), self.input.clone()).with_preserve_partitioning(true));
            SortPreservingMergeExec::new(sort_exprs, sort_exec).execute(0, context)?
        } else {
            self.input.execute(0, context)?
        };
        let output_schema = self.cache.eq_properties.schema().clone();
        let layout = writer_layout(&input_schema, &self.tag_fields, self.coordinate_system_zero_based);
        let stream = futures::stream::once(write_stream(input, layout, self.output_path.clone(), self.sort_on_write, self.device_id, output_schema.clone()));
        Ok(Box::pin(RecordBatchStreamAdapter::new(output_schema, stream)))
    }
}

/// The columns and the schema the C writer gets.  The reference serialises exactly the columns NAMED in `tag_fields`, in
/// that order, whatever metadata the input field carries: a column's SAM type is its `bio.bam.tag.type` metadata, `Z` when
/// there is none, a name that is not two bytes long is skipped, and a tag column of the input that is not listed is not
/// written (`build_tag_data`, bio-format-core/src/sam_tag_io.rs:109-147).  The C writer writes every field that carries
/// `bio.bam.tag.tag` in schema order and takes the coordinate system from `bio.coordinate_system_zero_based`, so: listed
/// columns get that key (value = the column name, the tag the reference writes), unlisted ones lose it, the tag columns
/// follow the other columns in `tag_fields` order, and the schema-level flag is the table's.
pub(crate) struct WriterLayout {
    indices: Vec<usize>,
    schema: SchemaRef,
}
pub(crate) fn writer_layout(input_schema: &SchemaRef, tag_fields: &[String], coordinate_system_zero_based: bool) -> WriterLayout {
    const TAG_KEY: &str = "bio.bam.tag.tag";
    let listed = |name: &str| tag_fields.iter().any(|t| t == name) && name.as_bytes().len() == 2;
    let mut indices: Vec<usize> = Vec::with_capacity(input_schema.fields().len());
    let mut fields: Vec<Field> = Vec::with_capacity(input_schema.fields().len());
    for (i, f) in input_schema.fields().iter().enumerate() {
        if listed(f.name()) {
            continue; // placed behind the other columns, in tag_fields order
        }
        let mut md = f.metadata().clone();
        md.remove(TAG_KEY); // a tag column of the input that the table does not list is not written
        indices.push(i);
        fields.push(f.as_ref().clone().with_metadata(md));
    }
    for t in tag_fields {
        if t.as_bytes().len() != 2 {
            continue;
        }
        if let Ok(i) = input_schema.index_of(t) {
            let f = input_schema.field(i);
            let mut md = f.metadata().clone();
            md.insert(TAG_KEY.to_string(), t.clone()); // (the SAM type stays the field's own `bio.bam.tag.type`, `Z` when absent)
            indices.push(i);
            fields.push(f.clone().with_metadata(md));
        }
    }
    let mut md = input_schema.metadata().clone();
    md.insert("bio.coordinate_system_zero_based".to_string(), coordinate_system_zero_based.to_string());
    WriterLayout { indices, schema: Arc::new(Schema::new_with_metadata(fields, md)) }
}

/// write_bam_stream (write_exec.rs:281-336): open, every batch, finish, one row with the count.
async fn write_stream(
    mut input: SendableRecordBatchStream,
    layout: WriterLayout,
    output_path: String,
    sort_on_write: bool,
    device_id: i32,
    output_schema: SchemaRef,
) -> datafusion::common::Result<RecordBatch> {
    let path = cstring(&output_path)?;
    let c_schema = FFI_ArrowSchema::try_from(layout.schema.as_ref()).map_err(|e| DataFusionError::ArrowError(Box::new(e), None))?;
    let mut raw: *mut ffi::bioscan_bam_writer = std::ptr::null_mut();
    check(unsafe { ffi::bioscan_bam_writer_open_schema(path.as_ptr(), &c_schema, sort_on_write as i32, device_id, &mut raw) })?;
    let writer = WriterHandle(raw);
    while let Some(batch) = input.next().await {
        let batch = batch?;
        if batch.num_rows() == 0 {
            continue;
        }
        // the writer's columns under the writer's schema (see writer_layout)
        let batch = batch.project(&layout.indices).and_then(|b| b.with_schema(layout.schema.clone())).map_err(|e| DataFusionError::ArrowError(Box::new(e), None))?;
        // a RecordBatch crosses the C Data Interface as a struct array (what bioscan_next hands out in the other direction)
        let data = StructArray::from(batch).into_data();
        let (c_array, c_batch_schema) = to_ffi(&data).map_err(|e| DataFusionError::ArrowError(Box::new(e), None))?;
        check(unsafe { ffi::bioscan_bam_writer_write(writer.0, &c_array, &c_batch_schema) })?;
    }
    let (mut n_records, mut n_members, mut n_bytes) = (0u64, 0u64, 0u64);
    check(unsafe { ffi::bioscan_bam_writer_finish(writer.0, &mut n_records, &mut n_members, &mut n_bytes) })?;
    log::debug!("wrote {n_records} records in {n_members} BGZF members ({n_bytes} bytes) to {output_path}");
    RecordBatch::try_new(output_schema, vec![Arc::new(UInt64Array::from(vec![n_records]))]).map_err(|e| DataFusionError::ArrowError(Box::new(e), None))
}
