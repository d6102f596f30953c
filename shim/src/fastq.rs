//! `FastqTableProvider` (bio-format-fastq/src/table_provider.rs:49-151) over `bioscan_fastq_open`.
use crate::ObjectStorageOptions;
use crate::exec::{BioscanExec, import_schema};
use crate::ffi;
use crate::handles::{ProviderHandle, check, cstring};
use arrow::datatypes::SchemaRef;
use async_trait::async_trait;
use datafusion::catalog::{Session, TableProvider};
use datafusion::datasource::TableType;
use datafusion::logical_expr::Expr;
use datafusion::physical_plan::ExecutionPlan;
use datafusion::physical_plan::execution_plan::EmissionType;
use std::any::Any;
use std::sync::Arc;

pub struct FastqTableProvider {
    provider: Arc<ProviderHandle>,
    schema: SchemaRef,
}

impl std::fmt::Debug for FastqTableProvider {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("FastqTableProvider").field("schema", &self.schema).finish()
    }
}

impl FastqTableProvider {
    pub fn new(file_path: String, object_storage_options: Option<ObjectStorageOptions>) -> datafusion::common::Result<Self> {
        let _ = object_storage_options;
        Self::new_on_device(file_path, 0)
    }

    pub fn new_on_device(file_path: String, device_id: i32) -> datafusion::common::Result<Self> {
        let path = cstring(&file_path)?;
        let mut raw: *mut ffi::bioscan_provider = std::ptr::null_mut();
        check(unsafe { ffi::bioscan_fastq_open(path.as_ptr(), device_id, &mut raw) })?;
        let provider = Arc::new(ProviderHandle(raw));
        let schema = import_schema(|s| unsafe { ffi::bioscan_schema(provider.0, s) })?;
        Ok(Self { provider, schema })
    }
}

#[async_trait]
impl TableProvider for FastqTableProvider {
    fn as_any(&self) -> &dyn Any {
        self
    }
    fn schema(&self) -> SchemaRef {
        self.schema.clone()
    }
    fn table_type(&self) -> TableType {
        TableType::Base
    }
    // no supports_filters_pushdown: the reference ignores filters (`_filters`, table_provider.rs:94)
    async fn scan(
        &self,
        state: &dyn Session,
        projection: Option<&Vec<usize>>,
        _filters: &[Expr],
        limit: Option<usize>,
    ) -> datafusion::common::Result<Arc<dyn ExecutionPlan>> {
        BioscanExec::plan("FastqExec", &self.provider, projection, &[], limit, state.config().target_partitions(), &[0],
                          EmissionType::Final)
    }
}
