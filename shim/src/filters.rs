//! `Expr` -> `bioscan_filter`: the shapes the reference can push down (bio-format-core/src/genomic_filter.rs:143-331,
//! record_filter.rs:40-356): `column <op> literal`, `column [NOT] BETWEEN lo AND hi`, `column [NOT] IN (...)`.
//! DataFusion hands `scan` the conjuncts of the WHERE clause one by one; an expression of another shape is simply not
//! sent (the library answers `Unsupported` for it and DataFusion evaluates it above the scan).
use crate::ffi;
use datafusion::logical_expr::{BinaryExpr, Between, Expr, Operator, expr::InList};
use datafusion::scalar::ScalarValue;
use std::ffi::CString;

/// Owns the strings and literal arrays a `[bioscan_filter]` points into.
pub(crate) struct FilterSet {
    pub raw: Vec<ffi::bioscan_filter>,
    /// index of the source expression of every entry of `raw`
    pub source: Vec<usize>,
    _columns: Vec<CString>,
    _strings: Vec<CString>,
    _literals: Vec<Vec<ffi::bioscan_literal>>,
}

fn literal(v: &ScalarValue, strings: &mut Vec<CString>) -> Option<ffi::bioscan_literal> {
    let mut l = ffi::bioscan_literal { kind: ffi::BIOSCAN_LIT_NULL, i: 0, f: 0.0, s: std::ptr::null() };
    match v {
        v if v.is_null() => {}
        ScalarValue::Int8(Some(x)) => { l.kind = ffi::BIOSCAN_LIT_INT; l.i = *x as i64 }
        ScalarValue::Int16(Some(x)) => { l.kind = ffi::BIOSCAN_LIT_INT; l.i = *x as i64 }
        ScalarValue::Int32(Some(x)) => { l.kind = ffi::BIOSCAN_LIT_INT; l.i = *x as i64 }
        ScalarValue::Int64(Some(x)) => { l.kind = ffi::BIOSCAN_LIT_INT; l.i = *x }
        ScalarValue::UInt8(Some(x)) => { l.kind = ffi::BIOSCAN_LIT_INT; l.i = *x as i64 }
        ScalarValue::UInt16(Some(x)) => { l.kind = ffi::BIOSCAN_LIT_INT; l.i = *x as i64 }
        ScalarValue::UInt32(Some(x)) => { l.kind = ffi::BIOSCAN_LIT_INT; l.i = *x as i64 }
        ScalarValue::UInt64(Some(x)) => { l.kind = ffi::BIOSCAN_LIT_INT; l.i = i64::try_from(*x).ok()? }
        ScalarValue::Float32(Some(x)) => { l.kind = ffi::BIOSCAN_LIT_FLOAT; l.f = *x as f64 }
        ScalarValue::Float64(Some(x)) => { l.kind = ffi::BIOSCAN_LIT_FLOAT; l.f = *x }
        ScalarValue::Utf8(Some(s)) | ScalarValue::LargeUtf8(Some(s)) | ScalarValue::Utf8View(Some(s)) => {
            let c = CString::new(s.as_str()).ok()?;
            l.kind = ffi::BIOSCAN_LIT_STR;
            l.s = c.as_ptr(); // the heap buffer of a CString does not move when the CString is moved into `strings`
            strings.push(c);
        }
        _ => return None,
    }
    Some(l)
}

fn column_name(e: &Expr) -> Option<&str> {
    match e {
        Expr::Column(c) => Some(c.name.as_str()),
        Expr::Cast(c) => column_name(&c.expr),
        Expr::TryCast(c) => column_name(&c.expr),
        _ => None,
    }
}

fn scalar(e: &Expr) -> Option<&ScalarValue> {
    match e {
        Expr::Literal(v, _) => Some(v),
        Expr::Cast(c) => scalar(&c.expr),
        Expr::TryCast(c) => scalar(&c.expr),
        _ => None,
    }
}

fn comparison(op: Operator, flipped: bool) -> Option<i32> {
    Some(match (op, flipped) {
        (Operator::Eq, _) => ffi::BIOSCAN_OP_EQ,
        (Operator::NotEq, _) => ffi::BIOSCAN_OP_NE,
        (Operator::Lt, false) | (Operator::Gt, true) => ffi::BIOSCAN_OP_LT,
        (Operator::LtEq, false) | (Operator::GtEq, true) => ffi::BIOSCAN_OP_LE,
        (Operator::Gt, false) | (Operator::Lt, true) => ffi::BIOSCAN_OP_GT,
        (Operator::GtEq, false) | (Operator::LtEq, true) => ffi::BIOSCAN_OP_GE,
        _ => return None,
    })
}

impl FilterSet {
    pub fn new(exprs: &[&Expr]) -> Self {
        let mut set = FilterSet { raw: vec![], source: vec![], _columns: vec![], _strings: vec![], _literals: vec![] };
        for (idx, e) in exprs.iter().enumerate() {
            let parsed: Option<(&str, i32, Vec<ffi::bioscan_literal>)> = match e {
                Expr::BinaryExpr(BinaryExpr { left, op, right }) => {
                    if let (Some(c), Some(v)) = (column_name(left), scalar(right)) {
                        comparison(*op, false).and_then(|o| literal(v, &mut set._strings).map(|l| (c, o, vec![l])))
                    } else if let (Some(v), Some(c)) = (scalar(left), column_name(right)) {
                        comparison(*op, true).and_then(|o| literal(v, &mut set._strings).map(|l| (c, o, vec![l])))
                    } else {
                        None
                    }
                }
                Expr::Between(Between { expr, negated, low, high }) => match (column_name(expr), scalar(low), scalar(high)) {
                    (Some(c), Some(lo), Some(hi)) => {
                        match (literal(lo, &mut set._strings), literal(hi, &mut set._strings)) {
                            (Some(a), Some(b)) => {
                                Some((c, if *negated { ffi::BIOSCAN_OP_NOT_BETWEEN } else { ffi::BIOSCAN_OP_BETWEEN }, vec![a, b]))
                            }
                            _ => None,
                        }
                    }
                    _ => None,
                },
                Expr::InList(InList { expr, list, negated }) => column_name(expr).and_then(|c| {
                    let mut vals = Vec::with_capacity(list.len());
                    for item in list {
                        vals.push(literal(scalar(item)?, &mut set._strings)?);
                    }
                    Some((c, if *negated { ffi::BIOSCAN_OP_NOT_IN } else { ffi::BIOSCAN_OP_IN }, vals))
                }),
                _ => None,
            };
            if let Some((col, op, vals)) = parsed {
                let Ok(col) = CString::new(col) else { continue };
                set._literals.push(vals);
                let vals = set._literals.last().unwrap();
                set.raw.push(ffi::bioscan_filter { column: col.as_ptr(), op, values: vals.as_ptr(), n_values: vals.len() as i32 });
                set._columns.push(col);
                set.source.push(idx);
            }
        }
        set
    }
}
