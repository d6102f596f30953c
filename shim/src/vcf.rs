//! `VcfTableProvider` (bio-format-vcf/src/table_provider.rs:752-1097, 1203-1462) over `bioscan_vcf_open`.
use crate::ObjectStorageOptions;
use crate::exec::{BioscanExec, import_schema, pushdown};
use crate::ffi;
use crate::handles::{ProviderHandle, check, cstring};
use arrow::datatypes::SchemaRef;
use async_trait::async_trait;
use datafusion::catalog::{Session, TableProvider};
use datafusion::datasource::TableType;
use datafusion::logical_expr::{Expr, TableProviderFilterPushDown};
use datafusion::physical_plan::ExecutionPlan;
use datafusion::physical_plan::execution_plan::EmissionType;
use std::any::Any;
use std::ffi::CString;
use std::os::raw::c_char;
use std::sync::Arc;

pub struct VcfTableProvider {
    provider: Arc<ProviderHandle>,
    schema: SchemaRef,
}

impl std::fmt::Debug for VcfTableProvider {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("VcfTableProvider").field("schema", &self.schema).finish()
    }
}

struct StrList {
    _own: Vec<CString>,
    ptrs: Vec<*const c_char>,
}
fn str_list(v: &Option<Vec<String>>) -> datafusion::common::Result<Option<StrList>> {
    match v {
        None => Ok(None),
        Some(items) => {
            let own: Vec<CString> = items.iter().map(|s| cstring(s)).collect::<Result<_, _>>()?;
            let ptrs = own.iter().map(|c| c.as_ptr()).collect();
            Ok(Some(StrList { _own: own, ptrs }))
        }
    }
}

impl VcfTableProvider {
    pub fn new(
        file_path: String,
        info_fields: Option<Vec<String>>,
        format_fields: Option<Vec<String>>,
        object_storage_options: Option<ObjectStorageOptions>,
        coordinate_system_zero_based: bool,
    ) -> datafusion::common::Result<Self> {
        Self::new_with_samples(file_path, info_fields, format_fields, None, object_storage_options, coordinate_system_zero_based)
    }

    /// `None` = every INFO / FORMAT tag of the header, every sample; `Some(vec![])` = none (table_provider.rs:849-1097).
    pub fn new_with_samples(
        file_path: String,
        info_fields: Option<Vec<String>>,
        format_fields: Option<Vec<String>>,
        samples_to_include: Option<Vec<String>>,
        object_storage_options: Option<ObjectStorageOptions>,
        coordinate_system_zero_based: bool,
    ) -> datafusion::common::Result<Self> {
        let _ = object_storage_options;
        Self::open(file_path, info_fields, format_fields, samples_to_include, coordinate_system_zero_based, None, 0)
    }

    pub fn open(
        file_path: String,
        info_fields: Option<Vec<String>>,
        format_fields: Option<Vec<String>>,
        samples_to_include: Option<Vec<String>>,
        coordinate_system_zero_based: bool,
        explicit_index_path: Option<String>,
        device_id: i32,
    ) -> datafusion::common::Result<Self> {
        let path = cstring(&file_path)?;
        let info = str_list(&info_fields)?;
        let format = str_list(&format_fields)?;
        let samples = str_list(&samples_to_include)?;
        let index = explicit_index_path.as_ref().map(|s| cstring(s)).transpose()?;
        let mut o: ffi::bioscan_vcf_options = unsafe { std::mem::zeroed() };
        unsafe { ffi::bioscan_vcf_options_default(&mut o) };
        o.device_id = device_id;
        o.coordinate_system_zero_based = coordinate_system_zero_based as i32;
        let dangling = std::ptr::NonNull::<*const c_char>::dangling().as_ptr() as *const *const c_char;
        if let Some(l) = &info {
            o.has_info_fields = 1;
            o.info_fields = if l.ptrs.is_empty() { dangling } else { l.ptrs.as_ptr() };
            o.n_info_fields = l.ptrs.len() as i32;
        }
        if let Some(l) = &format {
            o.has_format_fields = 1;
            o.format_fields = if l.ptrs.is_empty() { dangling } else { l.ptrs.as_ptr() };
            o.n_format_fields = l.ptrs.len() as i32;
        }
        if let Some(l) = &samples {
            o.has_samples = 1;
            o.samples = if l.ptrs.is_empty() { dangling } else { l.ptrs.as_ptr() };
            o.n_samples = l.ptrs.len() as i32;
        }
        o.index_path = index.as_ref().map(|c| c.as_ptr()).unwrap_or(std::ptr::null());
        let mut raw: *mut ffi::bioscan_provider = std::ptr::null_mut();
        check(unsafe { ffi::bioscan_vcf_open(path.as_ptr(), &o, &mut raw) })?;
        let provider = Arc::new(ProviderHandle(raw));
        let schema = import_schema(|s| unsafe { ffi::bioscan_schema(provider.0, s) })?;
        Ok(Self { provider, schema })
    }
}

#[async_trait]
impl TableProvider for VcfTableProvider {
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
        pushdown(&self.provider, filters)
    }
    async fn scan(
        &self,
        state: &dyn Session,
        projection: Option<&Vec<usize>>,
        filters: &[Expr],
        limit: Option<usize>,
    ) -> datafusion::common::Result<Arc<dyn ExecutionPlan>> {
        // VcfExec emits incrementally (physical_exec.rs:2560-2580)
        BioscanExec::plan("VCFExec", &self.provider, projection, filters, limit, state.config().target_partitions(), &[0],
                          EmissionType::Incremental)
    }
}
