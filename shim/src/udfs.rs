//! The list UDFs of bio-format-vcf/src/udfs.rs on the GPU: `list_avg` (:67-110), `list_gte` (:606-650), `list_lte`,
//! `list_and` (:765-850), `vcf_set_gts` (:857-953), `vcf_an` / `vcf_ac` / `vcf_af` (:161-552).  Arrow C Data in, Arrow C Data out;
//! signatures and NULL rules are the reference's.
use crate::ffi;
use crate::handles::{check, cstring};
use arrow::array::{Array, ArrayRef, make_array};
use arrow::datatypes::{DataType, Field};
use arrow::ffi::{FFI_ArrowArray, FFI_ArrowSchema, from_ffi, to_ffi};
use datafusion::common::{DataFusionError, Result};
use datafusion::logical_expr::TypeSignature::Exact;
use datafusion::logical_expr::{ColumnarValue, ScalarFunctionArgs, ScalarUDF, ScalarUDFImpl, Signature, Volatility};
use datafusion::scalar::ScalarValue;
use std::any::Any;
use std::sync::Arc;

const DEVICE: i32 = 0;

fn list_of(t: DataType) -> DataType {
    DataType::List(Arc::new(Field::new("item", t, true)))
}
fn arrow_err(e: arrow::error::ArrowError) -> DataFusionError {
    DataFusionError::ArrowError(Box::new(e), None)
}
fn export(a: &ArrayRef) -> Result<(FFI_ArrowArray, FFI_ArrowSchema)> {
    to_ffi(&a.to_data()).map_err(arrow_err)
}
fn import(out: FFI_ArrowArray, schema: FFI_ArrowSchema) -> Result<ColumnarValue> {
    let data = unsafe { from_ffi(out, &schema) }.map_err(arrow_err)?;
    Ok(ColumnarValue::Array(make_array(data)))
}
fn threshold(v: &ColumnarValue) -> Result<f64> {
    match v {
        ColumnarValue::Scalar(ScalarValue::Int32(Some(x))) => Ok(*x as f64),
        ColumnarValue::Scalar(ScalarValue::Int64(Some(x))) => Ok(*x as f64),
        ColumnarValue::Scalar(ScalarValue::Float32(Some(x))) => Ok(*x as f64),
        ColumnarValue::Scalar(ScalarValue::Float64(Some(x))) => Ok(*x),
        other => Err(DataFusionError::Execution(format!("list comparison threshold must be a numeric scalar, got {other:?}"))),
    }
}

macro_rules! udf_boilerplate {
    ($name:literal) => {
        fn as_any(&self) -> &dyn Any {
            self
        }
        fn name(&self) -> &str {
            $name
        }
        fn signature(&self) -> &Signature {
            &self.signature
        }
    };
}

#[derive(Debug, PartialEq, Eq, Hash)]
struct ListAvg {
    signature: Signature,
}
impl ScalarUDFImpl for ListAvg {
    udf_boilerplate!("list_avg");
    fn return_type(&self, _: &[DataType]) -> Result<DataType> {
        Ok(DataType::Float64)
    }
    fn invoke_with_args(&self, args: ScalarFunctionArgs) -> Result<ColumnarValue> {
        let (a, s) = export(&args.args[0].clone().into_array(1)?)?;
        let (mut oa, mut os) = (FFI_ArrowArray::empty(), FFI_ArrowSchema::empty());
        check(unsafe { ffi::bioscan_udf_list_avg(&a, &s, DEVICE, &mut oa, &mut os) })?;
        import(oa, os)
    }
}

#[derive(Debug, PartialEq, Eq, Hash)]
struct ListCmp {
    signature: Signature,
    lte: bool,
}
impl ScalarUDFImpl for ListCmp {
    fn as_any(&self) -> &dyn Any {
        self
    }
    fn name(&self) -> &str {
        if self.lte { "list_lte" } else { "list_gte" }
    }
    fn signature(&self) -> &Signature {
        &self.signature
    }
    fn return_type(&self, _: &[DataType]) -> Result<DataType> {
        Ok(list_of(DataType::Boolean))
    }
    fn invoke_with_args(&self, args: ScalarFunctionArgs) -> Result<ColumnarValue> {
        let (a, s) = export(&args.args[0].clone().into_array(1)?)?;
        let t = threshold(&args.args[1])?;
        let (mut oa, mut os) = (FFI_ArrowArray::empty(), FFI_ArrowSchema::empty());
        check(unsafe { ffi::bioscan_udf_list_cmp(&a, &s, self.lte as i32, t, DEVICE, &mut oa, &mut os) })?;
        import(oa, os)
    }
}

#[derive(Debug, PartialEq, Eq, Hash)]
struct ListAnd {
    signature: Signature,
}
impl ScalarUDFImpl for ListAnd {
    udf_boilerplate!("list_and");
    fn return_type(&self, _: &[DataType]) -> Result<DataType> {
        Ok(list_of(DataType::Boolean))
    }
    fn invoke_with_args(&self, args: ScalarFunctionArgs) -> Result<ColumnarValue> {
        let n = args.number_rows;
        let (a, sa) = export(&args.args[0].clone().into_array(n)?)?;
        let (b, sb) = export(&args.args[1].clone().into_array(n)?)?;
        let (mut oa, mut os) = (FFI_ArrowArray::empty(), FFI_ArrowSchema::empty());
        check(unsafe { ffi::bioscan_udf_list_and(&a, &sa, &b, &sb, DEVICE, &mut oa, &mut os) })?;
        import(oa, os)
    }
}

#[derive(Debug, PartialEq, Eq, Hash)]
struct VcfSetGts {
    signature: Signature,
}
impl ScalarUDFImpl for VcfSetGts {
    udf_boilerplate!("vcf_set_gts");
    fn return_type(&self, _: &[DataType]) -> Result<DataType> {
        Ok(list_of(DataType::Utf8))
    }
    fn invoke_with_args(&self, args: ScalarFunctionArgs) -> Result<ColumnarValue> {
        let n = args.number_rows;
        let (g, sg) = export(&args.args[0].clone().into_array(n)?)?;
        let (m, sm) = export(&args.args[1].clone().into_array(n)?)?;
        let repl = match &args.args[2] {
            ColumnarValue::Scalar(ScalarValue::Utf8(Some(s))) | ColumnarValue::Scalar(ScalarValue::LargeUtf8(Some(s))) => cstring(s)?,
            other => return Err(DataFusionError::Execution(format!("vcf_set_gts replacement must be a string scalar, got {other:?}"))),
        };
        let (mut oa, mut os) = (FFI_ArrowArray::empty(), FFI_ArrowSchema::empty());
        check(unsafe { ffi::bioscan_udf_vcf_set_gts(&g, &sg, &m, &sm, repl.as_ptr(), DEVICE, &mut oa, &mut os) })?;
        import(oa, os)
    }
}

fn list_sig_1() -> Signature {
    Signature::one_of(vec![Exact(vec![list_of(DataType::Int32)]), Exact(vec![list_of(DataType::Float32)])], Volatility::Immutable)
}
fn list_sig_cmp() -> Signature {
    Signature::one_of(
        vec![
            Exact(vec![list_of(DataType::Int32), DataType::Int32]),
            Exact(vec![list_of(DataType::Int32), DataType::Int64]),
            Exact(vec![list_of(DataType::Float32), DataType::Float32]),
            Exact(vec![list_of(DataType::Float32), DataType::Float64]),
        ],
        Volatility::Immutable,
    )
}

pub fn list_avg_udf() -> ScalarUDF {
    ScalarUDF::from(ListAvg { signature: list_sig_1() })
}
pub fn list_gte_udf() -> ScalarUDF {
    ScalarUDF::from(ListCmp { signature: list_sig_cmp(), lte: false })
}
pub fn list_lte_udf() -> ScalarUDF {
    ScalarUDF::from(ListCmp { signature: list_sig_cmp(), lte: true })
}
pub fn list_and_udf() -> ScalarUDF {
    ScalarUDF::from(ListAnd {
        signature: Signature::exact(vec![list_of(DataType::Boolean), list_of(DataType::Boolean)], Volatility::Immutable),
    })
}
pub fn vcf_set_gts_udf() -> ScalarUDF {
    ScalarUDF::from(VcfSetGts {
        signature: Signature::exact(vec![list_of(DataType::Utf8), list_of(DataType::Boolean), DataType::Utf8], Volatility::Immutable),
    })
}

/// `vcf_an` / `vcf_ac` / `vcf_af` (bio-format-vcf/src/udfs.rs:161-552) over `bioscan_udf_vcf_allele_stats`; the optional second
/// argument of AC / AF is the provider's pipe-separated ALT column (a scalar is broadcast, as `into_array(list.len())` does).
#[derive(Debug, PartialEq, Eq, Hash)]
struct AlleleStat {
    signature: Signature,
    which: i32,
}
impl ScalarUDFImpl for AlleleStat {
    fn as_any(&self) -> &dyn Any {
        self
    }
    fn name(&self) -> &str {
        match self.which {
            0 => "vcf_an",
            1 => "vcf_ac",
            _ => "vcf_af",
        }
    }
    fn signature(&self) -> &Signature {
        &self.signature
    }
    fn return_type(&self, _: &[DataType]) -> Result<DataType> {
        Ok(match self.which {
            0 => DataType::Int32,
            1 => list_of(DataType::Int32),
            _ => list_of(DataType::Float64),
        })
    }
    fn invoke_with_args(&self, args: ScalarFunctionArgs) -> Result<ColumnarValue> {
        let gt = args.args[0].clone().into_array(1)?;
        let (ga, gs) = export(&gt)?;
        let alt = if args.args.len() > 1 { Some(export(&args.args[1].clone().into_array(gt.len())?)?) } else { None };
        let (ap, sp) = match &alt {
            Some((a, s)) => (a as *const FFI_ArrowArray, s as *const FFI_ArrowSchema),
            None => (std::ptr::null(), std::ptr::null()),
        };
        let (mut oa, mut os) = (FFI_ArrowArray::empty(), FFI_ArrowSchema::empty());
        check(unsafe { ffi::bioscan_udf_vcf_allele_stats(&ga, &gs, ap, sp, self.which, DEVICE, &mut oa, &mut os) })?;
        import(oa, os)
    }
}
fn allele_sig(two_arg: bool) -> Signature {
    if two_arg {
        Signature::one_of(vec![Exact(vec![list_of(DataType::Utf8)]), Exact(vec![list_of(DataType::Utf8), DataType::Utf8])], Volatility::Immutable)
    } else {
        Signature::exact(vec![list_of(DataType::Utf8)], Volatility::Immutable)
    }
}
pub fn vcf_an_udf() -> ScalarUDF {
    ScalarUDF::from(AlleleStat { signature: allele_sig(false), which: 0 })
}
pub fn vcf_ac_udf() -> ScalarUDF {
    ScalarUDF::from(AlleleStat { signature: allele_sig(true), which: 1 })
}
pub fn vcf_af_udf() -> ScalarUDF {
    ScalarUDF::from(AlleleStat { signature: allele_sig(true), which: 2 })
}

/// `register_vcf_udfs` (bio-format-vcf/src/udfs.rs:976-986): every UDF of the reference under its name.
pub fn register_vcf_udfs(ctx: &datafusion::prelude::SessionContext) {
    ctx.register_udf(vcf_an_udf());
    ctx.register_udf(vcf_ac_udf());
    ctx.register_udf(vcf_af_udf());
    ctx.register_udf(list_avg_udf());
    ctx.register_udf(list_gte_udf());
    ctx.register_udf(list_lte_udf());
    ctx.register_udf(list_and_udf());
    ctx.register_udf(vcf_set_gts_udf());
}
