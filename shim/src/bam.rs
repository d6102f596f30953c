//! `BamTableProvider` (bio-format-bam/src/table_provider.rs:381-529, 639-662, 927-1178) over `bioscan_bam_open` (read side)
//! and `bioscan_bam_writer_open_schema` (write side, through `write::BamWriteExec`).
use crate::ObjectStorageOptions;
use crate::exec::{BioscanExec, import_schema, pushdown};
use crate::ffi;
use crate::handles::{ProviderHandle, check, cstring};
use crate::write::BamWriteExec;
use arrow::datatypes::SchemaRef;
use async_trait::async_trait;
use datafusion::catalog::{Session, TableProvider};
use datafusion::datasource::TableType;
use datafusion::logical_expr::dml::InsertOp;
use datafusion::logical_expr::{Expr, TableProviderFilterPushDown};
use datafusion::physical_plan::ExecutionPlan;
use datafusion::physical_plan::execution_plan::EmissionType;
use std::any::Any;
use std::ffi::CString;
use std::sync::Arc;

/// What the GPU path adds to the reference constructor's arguments.
#[derive(Debug, Clone)]
pub struct BamOptions {
    /// GPUs of this node.  One id: every partition runs there.  Several: the plan's partitions are dealt to them in
    /// contiguous runs in plan order (`bioscan_scan_devices`), each GPU holding only its share of the file.
    pub device_ids: Vec<i32>,
    /// `Some(path)`: this index; `Some("")`: no index; `None`: discover `<file>.bai` / `<stem>.bai`.
    pub index_path: Option<String>,
    /// BGZF members per pipeline chunk of a stream (bounds HBM and host memory per `execute`); 0 = default.
    pub chunk_members: i32,
}

impl Default for BamOptions {
    fn default() -> Self {
        Self { device_ids: vec![0], index_path: None, chunk_members: 0 }
    }
}

pub struct BamTableProvider {
    /// `None` for a provider made by `new_for_write`: there is no file to scan yet.
    provider: Option<Arc<ProviderHandle>>,
    schema: SchemaRef,
    options: BamOptions,
    /// write side (`new_for_write`, table_provider.rs:639-662)
    output_path: Option<String>,
    write_tag_fields: Option<Vec<String>>,
    sort_on_write: bool,
}

impl std::fmt::Debug for BamTableProvider {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("BamTableProvider").field("schema", &self.schema).field("options", &self.options).finish()
    }
}

impl BamTableProvider {
    /// The reference constructor, argument for argument (`object_storage_options` is accepted and ignored: local files).
    #[allow(clippy::too_many_arguments)]
    pub async fn new(
        file_path: String,
        object_storage_options: Option<ObjectStorageOptions>,
        coordinate_system_zero_based: bool,
        tag_fields: Option<Vec<String>>,
        binary_cigar: bool,
        infer_tag_types: bool,
        infer_tag_sample_size: usize,
        tag_type_hints: Option<Vec<String>>,
    ) -> datafusion::common::Result<Self> {
        let _ = object_storage_options;
        Self::new_with_options(file_path, coordinate_system_zero_based, tag_fields, binary_cigar, infer_tag_types,
                               infer_tag_sample_size, tag_type_hints, BamOptions::default())
    }

    #[allow(clippy::too_many_arguments)]
    pub fn new_with_options(
        file_path: String,
        coordinate_system_zero_based: bool,
        tag_fields: Option<Vec<String>>,
        binary_cigar: bool,
        infer_tag_types: bool,
        infer_tag_sample_size: usize,
        tag_type_hints: Option<Vec<String>>,
        options: BamOptions,
    ) -> datafusion::common::Result<Self> {
        let path = cstring(&file_path)?;
        let tags: Option<Vec<CString>> = tag_fields.as_ref().map(|v| v.iter().map(|s| cstring(s)).collect()).transpose()?;
        let tag_ptrs: Option<Vec<*const std::os::raw::c_char>> = tags.as_ref().map(|v| v.iter().map(|c| c.as_ptr()).collect());
        let hints: Option<Vec<CString>> = tag_type_hints.as_ref().map(|v| v.iter().map(|s| cstring(s)).collect()).transpose()?;
        let hint_ptrs: Option<Vec<*const std::os::raw::c_char>> = hints.as_ref().map(|v| v.iter().map(|c| c.as_ptr()).collect());
        let index = options.index_path.as_ref().map(|s| cstring(s)).transpose()?;
        let mut o: ffi::bioscan_bam_options = unsafe { std::mem::zeroed() };
        unsafe { ffi::bioscan_bam_options_default(&mut o) };
        o.coordinate_system_zero_based = coordinate_system_zero_based as i32;
        if let Some(p) = &tag_ptrs {
            // an empty selection is still a selection: a dangling non-null pointer with n = 0
            o.tag_fields = if p.is_empty() { std::ptr::NonNull::dangling().as_ptr() } else { p.as_ptr() };
            o.n_tag_fields = p.len() as i32;
        }
        o.binary_cigar = binary_cigar as i32;
        o.infer_tag_types = infer_tag_types as i32;
        o.infer_tag_sample_size = infer_tag_sample_size as i32;
        if let Some(p) = &hint_ptrs {
            o.tag_type_hints = p.as_ptr();
            o.n_tag_type_hints = p.len() as i32;
        }
        o.index_path = index.as_ref().map(|c| c.as_ptr()).unwrap_or(std::ptr::null());
        o.device_id = *options.device_ids.first().unwrap_or(&0);
        o.chunk_members = options.chunk_members;
        let mut raw: *mut ffi::bioscan_provider = std::ptr::null_mut();
        check(unsafe { ffi::bioscan_bam_open(path.as_ptr(), &o, &mut raw) })?;
        let provider = Arc::new(ProviderHandle(raw));
        let schema = import_schema(|s| unsafe { ffi::bioscan_schema(provider.0, s) })?;
        Ok(Self { provider: Some(provider), schema, options, output_path: None, write_tag_fields: None, sort_on_write: false })
    }

    /// The reference's write constructor, argument for argument (table_provider.rs:639-662).  `coordinate_system_zero_based`
    /// is accepted and -- as in the reference's `insert_into`, which reads the schema's
    /// `bio.coordinate_system_zero_based` instead (:1131-1135) -- not used.
    pub fn new_for_write(
        output_path: String,
        schema: SchemaRef,
        tag_fields: Option<Vec<String>>,
        coordinate_system_zero_based: bool,
        sort_on_write: bool,
    ) -> Self {
        let _ = coordinate_system_zero_based;
        Self { provider: None, schema, options: BamOptions::default(), output_path: Some(output_path), write_tag_fields: tag_fields, sort_on_write }
    }

    fn read_side(&self) -> datafusion::common::Result<&Arc<ProviderHandle>> {
        self.provider.as_ref().ok_or_else(|| datafusion::common::DataFusionError::Execution("this BamTableProvider was created for writing: nothing to scan".to_string()))
    }
}

#[async_trait]
impl TableProvider for BamTableProvider {
    fn as_any(&self) -> &dyn Any {
        self
    }
    fn schema(&self) -> SchemaRef {
        self.schema.clone()
    }
    fn table_type(&self) -> TableType {
        TableType::Base
    }
    fn supports_filters_pushdown(&self, filters: &[&Expr]) -> datafusion::common::Result<Vec<TableProviderFilterPushDown>> {
        pushdown(self.read_side()?, filters)
    }
    async fn scan(
        &self,
        state: &dyn Session,
        projection: Option<&Vec<usize>>,
        filters: &[Expr],
        limit: Option<usize>,
    ) -> datafusion::common::Result<Arc<dyn ExecutionPlan>> {
        BioscanExec::plan("BamExec", self.read_side()?, projection, filters, limit, state.config().target_partitions(),
                          &self.options.device_ids, EmissionType::Final)
    }

    /// `INSERT OVERWRITE` (table_provider.rs:1117-1178): the tag columns are the schema's fields that carry
    /// `bio.bam.tag.tag` metadata plus the names given to `new_for_write`; the header's sort order follows `sort_on_write`.
    async fn insert_into(
        &self,
        _state: &dyn Session,
        input: Arc<dyn ExecutionPlan>,
        insert_op: InsertOp,
    ) -> datafusion::common::Result<Arc<dyn ExecutionPlan>> {
        if insert_op != InsertOp::Overwrite {
            return Err(datafusion::common::DataFusionError::NotImplemented("BAM insert_into only supports OVERWRITE mode".to_string()));
        }
        let path = match &self.output_path {
            Some(p) => p.clone(),
            None => return Err(datafusion::common::DataFusionError::Execution("this BamTableProvider was opened for reading: use new_for_write".to_string())),
        };
        let mut tag_fields: Vec<String> = self.schema.fields().iter().filter(|f| f.metadata().contains_key("bio.bam.tag.tag")).map(|f| f.name().clone()).collect();
        if let Some(explicit) = &self.write_tag_fields {
            for t in explicit {
                if !tag_fields.contains(t) {
                    tag_fields.push(t.clone());
                }
            }
        }
        // the coordinate system of the rows is the TABLE's (table_provider.rs:1131-1135), not the input plan's
        let zero_based = self.schema.metadata().get("bio.coordinate_system_zero_based").and_then(|s| s.parse::<bool>().ok()).unwrap_or(true);
        Ok(Arc::new(BamWriteExec::new(input, path, tag_fields, zero_based, self.sort_on_write, *self.options.device_ids.first().unwrap_or(&0))))
    }
}
