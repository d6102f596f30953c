"""The Rust shim (shim/) cannot be compiled in this image (no cargo / rustc), so the one thing that can silently rot --
its `extern "C"` block against include/bioscan.h -- is checked here: same functions, same parameter counts, and every
#[repr(C)] struct has the header's fields in the header's order."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", "", s, flags=re.S)


def _split_params(p):
    p = p.strip()
    if not p or p == "void":
        return []
    out, depth, cur = [], 0, ""
    for ch in p:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _c_functions(hdr):
    fns = {}
    for m in re.finditer(r"\b(bioscan_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        fns[m.group(1)] = len(_split_params(m.group(2)))
    return fns


def _rust_functions(src):
    block = src[src.index('unsafe extern "C" {'):]
    fns = {}
    for m in re.finditer(r"pub fn (bioscan_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", block, flags=re.S):
        fns[m.group(1)] = len(_split_params(m.group(2)))
    return fns


def _c_struct_fields(hdr, name):
    m = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), hdr, flags=re.S)
    assert m, name
    fields = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if decl:
            fields.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", decl)[0])
    return fields


def _rust_struct_fields(src, name):
    m = re.search(r"pub struct %s \{(.*?)\n\}" % name, src, flags=re.S)
    assert m, name
    return re.findall(r"pub ([a-z_0-9]+):", m.group(1))


def test_extern_block_matches_header():
    hdr = _strip_c_comments(open(os.path.join(ROOT, "include", "bioscan.h")).read())
    src = open(os.path.join(ROOT, "shim", "src", "ffi.rs")).read()
    c, r = _c_functions(hdr), _rust_functions(src)
    assert len(c) >= 35
    assert sorted(set(c) - set(r)) == [], "declared in bioscan.h, missing in shim/src/ffi.rs"
    assert sorted(set(r) - set(c)) == [], "declared in shim/src/ffi.rs, not in bioscan.h"
    for name, n in c.items():
        assert r[name] == n, (name, "parameters: header", n, "shim", r[name])


def test_repr_c_structs_match_header():
    hdr = _strip_c_comments(open(os.path.join(ROOT, "include", "bioscan.h")).read())
    src = open(os.path.join(ROOT, "shim", "src", "ffi.rs")).read()
    for name in ("bioscan_bam_options", "bioscan_vcf_options", "bioscan_udf_stats", "bioscan_literal", "bioscan_filter",
                 "bioscan_scan_stats"):
        assert _rust_struct_fields(src, name) == _c_struct_fields(hdr, name), name
    # the enum values the shim hard-codes
    ops = re.search(r"enum bioscan_filter_op \{(.*?)\}", hdr, flags=re.S).group(1)
    names = [x.strip().split("=")[0].strip() for x in ops.split(",") if x.strip()]
    for i, n in enumerate(names):
        assert re.search(r"pub const %s: i32 = %d;" % (n, i), src), n
    lits = re.search(r"enum bioscan_literal_kind \{(.*?)\}", hdr, flags=re.S).group(1)
    for i, n in enumerate(x.strip().split("=")[0].strip() for x in lits.split(",") if x.strip()):
        assert re.search(r"pub const %s: i32 = %d;" % (n, i), src), n


def test_every_provider_entry_point_is_used_by_the_shim():
    """The shim must route through the ABI, not around it: each provider / plan / stream entry point appears in the crate."""
    used = ""
    for f in os.listdir(os.path.join(ROOT, "shim", "src")):
        if f != "ffi.rs":
            used += open(os.path.join(ROOT, "shim", "src", f)).read()
    for sym in ("bioscan_bam_open", "bioscan_vcf_open", "bioscan_fastq_open", "bioscan_schema", "bioscan_supports_filters_pushdown",
                "bioscan_scan", "bioscan_scan_devices", "bioscan_plan_num_partitions", "bioscan_plan_schema", "bioscan_plan_display",
                "bioscan_execute", "bioscan_next", "bioscan_stream_close", "bioscan_plan_close", "bioscan_provider_close",
                "bioscan_last_error", "bioscan_udf_list_avg", "bioscan_udf_list_cmp", "bioscan_udf_list_and", "bioscan_udf_vcf_set_gts"):
        assert "ffi::" + sym in used, sym
