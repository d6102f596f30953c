//! Owning wrappers of the three opaque C handles + error plumbing.
use crate::ffi;
use datafusion::common::DataFusionError;
use std::ffi::{CStr, CString};

/// `bioscan_last_error()` of this thread as the error the reference yields (`DataFusionError::Execution`,
/// bio-format-bam/src/physical_exec.rs:567-571).
pub(crate) fn last_error() -> DataFusionError {
    let msg = unsafe {
        let p = ffi::bioscan_last_error();
        if p.is_null() { "unknown error".to_string() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
    };
    DataFusionError::Execution(msg)
}

pub(crate) fn check(rc: i32) -> datafusion::common::Result<()> {
    if rc == 0 { Ok(()) } else { Err(last_error()) }
}

pub(crate) fn cstring(s: &str) -> datafusion::common::Result<CString> {
    CString::new(s).map_err(|_| DataFusionError::Configuration(format!("string contains a NUL byte: {s:?}")))
}

/// `BamTableProvider` / `VcfTableProvider` / `FastqTableProvider` on the C side.  Immutable after open (bioscan.h:
/// every execute owns its stream and scratch), hence `Send + Sync`.
pub(crate) struct ProviderHandle(pub *mut ffi::bioscan_provider);
unsafe impl Send for ProviderHandle {}
unsafe impl Sync for ProviderHandle {}
impl Drop for ProviderHandle {
    fn drop(&mut self) {
        unsafe { ffi::bioscan_provider_close(self.0) }
    }
}

/// An `ExecutionPlan` on the C side; keeps its provider alive.
pub(crate) struct PlanHandle {
    pub raw: *mut ffi::bioscan_plan,
    pub _provider: std::sync::Arc<ProviderHandle>,
}
unsafe impl Send for PlanHandle {}
unsafe impl Sync for PlanHandle {}
impl Drop for PlanHandle {
    fn drop(&mut self) {
        unsafe { ffi::bioscan_plan_close(self.raw) }
    }
}

/// A `SendableRecordBatchStream` on the C side: used by one thread at a time (`Send`, not `Sync`), like the producer
/// closure of `sync_batch_stream` (bio-format-core/src/sync_stream.rs:34-43).
pub(crate) struct StreamHandle {
    pub raw: *mut ffi::bioscan_stream,
    pub _plan: std::sync::Arc<PlanHandle>,
}
unsafe impl Send for StreamHandle {}
impl Drop for StreamHandle {
    fn drop(&mut self) {
        unsafe { ffi::bioscan_stream_close(self.raw) }
    }
}
