"""Known-answer tests the reference itself holds for the scan path, transcribed as DATA (inputs + the values the
reference asserts) and run against the oracle and -- wherever a host-only hook exists -- the C++ planner of the product:

  bio-format-core/src/record_filter.rs:358-520      7 tests  (evaluate_record_filters, can_push_down_record_filter)
  bio-format-core/src/genomic_filter.rs:350-557    16 tests  (extract_genomic_regions, is_genomic_coordinate_filter)
  bio-format-core/src/alignment_utils.rs:818-1017           (CIGAR formatting, binary CIGAR encode / decode)
  bio-format-core/src/tag_registry.rs:794-909                (registry, type mapping, type hints, inference)
  bio-format-core/src/range_planning.rs:328-374              (in tests/test_cpu_host_logic.py)

The device side of the same vectors (k_row_flags, bioscan_supports_filters_pushdown, tag back-fill) is in
tests/test_gpu_reference_kats.py."""
import struct

import pyarrow as pa
import pytest

# ---- record_filter.rs:358-520 -------------------------------------------------------------------------------------
# TestRecord { chrom: "chr1", start: 1000, mapping_quality: 30 }: string field chrom, u32 fields start / mapping_quality
REC = {"chrom": "chr1", "start": 1000, "mapping_quality": 30}
RECORD_FILTER_KATS = [
    # (test name, filters, expected)
    ("test_evaluate_chrom_eq", [("chrom", "=", "chr1")], True),
    ("test_evaluate_chrom_eq/miss", [("chrom", "=", "chr2")], False),
    ("test_evaluate_numeric_gte", [("mapping_quality", ">=", 30)], True),
    ("test_evaluate_numeric_gte/fail", [("mapping_quality", ">=", 31)], False),
    ("test_empty_filters", [], True),
    ("test_in_list_filter", [("chrom", "in", ["chr1", "chr2"])], True),
    ("test_in_list_filter/miss", [("chrom", "in", ["chr2", "chr3"])], False),
]


@pytest.mark.parametrize("name,filters,want", RECORD_FILTER_KATS)
def test_record_filter_kats_on_oracle(oracle, name, filters, want):
    assert oracle.evaluate_record_filters(REC, filters, string_fields=("chrom",), num_fields=("start", "mapping_quality")) is want


def test_nullable_fields_do_not_pass_comparisons_or_negated_lists(oracle):
    # NullableNumericRecord { score: None } with is_null_field("score") == true (record_filter.rs:497-507)
    kw = dict(string_fields=(), num_fields=("score",), null_fields=("score",))
    assert oracle.evaluate_record_filters({"score": None}, [("score", "!=", 10.0)], **kw) is False
    assert oracle.evaluate_record_filters({"score": None}, [("score", "not in", [10.0])], **kw) is False
    # ... while a BAM record's missing value has no accessor at all and passes (storage.rs:469-494, record_filter.rs:108-110)
    assert oracle.evaluate_record_filters({"start": None}, [("start", "!=", 10)]) is True


def test_numeric_in_list_supports_f64_accessors(oracle):
    kw = dict(string_fields=(), num_fields=("score",))
    assert oracle.evaluate_record_filters({"score": 3.5}, [("score", "in", [1.5, 3.5])], **kw) is True
    assert oracle.evaluate_record_filters({"score": 3.5}, [("score", "in", [1.5, 2.5])], **kw) is False


def test_in_list_null_rules(oracle):
    # evaluate_in_list (record_filter.rs:182-199): a non-match in a list that holds NULL is UNKNOWN for IN and NOT IN alike
    assert oracle.evaluate_record_filters(REC, [("chrom", "in", ["chr2", None])]) is False
    assert oracle.evaluate_record_filters(REC, [("chrom", "not in", ["chr2", None])]) is False
    assert oracle.evaluate_record_filters(REC, [("chrom", "not in", ["chr2"])]) is True
    assert oracle.evaluate_record_filters(REC, [("chrom", "in", ["chr1", None])]) is True
    # a NULL literal in a comparison never passes (record_filter.rs:87)
    assert oracle.evaluate_record_filters(REC, [("start", "=", None)], num_fields=("start",)) is False


def test_can_push_down(oracle):
    # record_filter.rs:455-471 (LIKE has no representation in the filter struct: it is never offered to the library)
    assert oracle.can_push_down_record_filter(("chrom", "=", "chr1"))
    assert oracle.can_push_down_record_filter(("start", ">=", 1000))
    assert not oracle.can_push_down_record_filter(("chrom", "<", "chr1"))


# ---- genomic_filter.rs:350-557 ------------------------------------------------------------------------------------
# (test name, filters, zero_based) -> regions [(chrom, start, end)], unsatisfiable, number of residual (non-genomic) filters
GENOMIC_KATS = [
    ("test_extract_chrom_eq", [("chrom", "=", "chr1")], True, [("chr1", None, None)], False, 0),
    ("test_extract_chrom_in_list", [("chrom", "in", ["chr1", "chr2"])], True, [("chr1", None, None), ("chr2", None, None)], False, 0),
    ("test_extract_chrom_with_range_zero_based", [("chrom", "=", "chr1"), ("start", ">=", 999), ("end", "<=", 2000)], True,
     [("chr1", 1000, 2000)], False, 0),
    ("test_extract_chrom_with_range_one_based", [("chrom", "=", "chr1"), ("start", ">=", 1000), ("end", "<=", 2000)], False,
     [("chr1", 1000, 2000)], False, 0),
    ("test_extract_chrom_with_exact_start_bounds_region_zero_based", [("chrom", "=", "chr1"), ("start", "=", 999)], True,
     [("chr1", 1000, 1000)], False, 0),
    ("test_extract_chrom_with_exact_start_bounds_region_one_based", [("chrom", "=", "chr1"), ("start", "=", 1000)], False,
     [("chr1", 1000, 1000)], False, 0),
    ("test_extract_chrom_with_contradictory_start_bounds_skips_invalid_region",
     [("chrom", "=", "chr1"), ("start", "=", 1000), ("start", ">", 1000)], False, [], True, 0),
    ("test_extract_chrom_with_start_upper_bound_one_based", [("chrom", "=", "chr1"), ("start", ">=", 1000), ("start", "<=", 2000)], False,
     [("chr1", 1000, 2000)], False, 0),
    ("test_extract_chrom_with_start_upper_bound_zero_based", [("chrom", "=", "chr1"), ("start", ">=", 999), ("start", "<=", 1999)], True,
     [("chr1", 1000, 2000)], False, 0),
    ("test_extract_chrom_with_start_exclusive_upper_bound_one_based", [("chrom", "=", "chr1"), ("start", ">=", 1000), ("start", "<", 2000)],
     False, [("chr1", 1000, 1999)], False, 0),
    ("test_non_genomic_filter_becomes_residual", [("chrom", "=", "chr1"), ("mapping_quality", ">=", 30)], True, [("chr1", None, None)], False, 1),
    ("test_no_genomic_filters", [("mapping_quality", ">=", 30)], True, [], False, 1),
    ("test_between_start", [("start", "between", (999, 1999))], True, [], False, 0),
    ("test_between_with_chrom", [("chrom", "=", "chr1"), ("start", "between", (999, 1999))], True, [("chr1", 1000, 2000)], False, 0),
]


@pytest.mark.parametrize("name,filters,zero_based,regions,unsat,residual", GENOMIC_KATS)
def test_genomic_filter_kats_on_oracle(oracle, name, filters, zero_based, regions, unsat, residual):
    got, u = oracle.extract_genomic_regions(filters, zero_based)
    assert [(r.chrom, r.start, r.end) for r in got] == regions
    assert u is unsat


@pytest.mark.parametrize("name,filters,zero_based,regions,unsat,residual", GENOMIC_KATS)
def test_genomic_filter_kats_on_cpp_planner(pkg, name, filters, zero_based, regions, unsat, residual):
    want = ";".join(f"{c}:{'' if s is None else s}-{'' if e is None else e}" for c, s, e in regions)
    got = pkg.debug_extract_regions(filters, zero_based)
    head, u, g, r = got.split("|")
    assert head == want, got
    assert u == f"unsat={1 if unsat else 0}", got
    assert r == f"residual={residual}", got           # is_genomic_coordinate_filter splits the conjuncts the same way


def test_is_genomic_coordinate_filter(pkg):
    # genomic_filter.rs:521-530
    assert pkg.debug_extract_regions([("chrom", "=", "chr1")]).endswith("genomic=1|residual=0")
    assert pkg.debug_extract_regions([("start", ">=", 1000)]).endswith("genomic=1|residual=0")
    assert pkg.debug_extract_regions([("mapping_quality", ">=", 30)]).endswith("genomic=0|residual=1")


def test_build_full_scan_regions(oracle, golden):
    # genomic_filter.rs:512-519: one unbounded region per reference, in header order
    import os
    o = oracle.BamOracle(os.path.join(golden, "multi_chrom.bam"))
    parts, _ = o.scan(target_partitions=1)
    chroms = [r.chrom for p in parts for r in p.regions if not r.unmapped_tail]
    assert chroms == sorted(set(chroms), key=chroms.index) and set(chroms) <= set(o.hdr.ref_names)


# ---- alignment_utils.rs:818-1017 ----------------------------------------------------------------------------------
def _encode(ops):  # encode_cigar_ops_to_binary: u32 LE, len << 4 | op code (MIDNSHP=X)
    return b"".join(struct.pack("<I", (n << 4) | "MIDNSHP=X".index(k)) for n, k in ops)


def _packed(ops):  # the oracle's op representation: the BAM u32 (len << 4 | code)
    return [(n << 4) | "MIDNSHP=X".index(k) for n, k in ops]


def test_format_cigar_ops(oracle):
    # alignment_utils.rs:860-879: the buffer is cleared per call
    assert oracle.cigar_string(_packed([(50, "M")])) == "50M"
    assert oracle.cigar_string(_packed([(2, "H"), (100, "M")])) == "2H100M"
    assert oracle.cigar_string([]) == ""


def test_encode_decode_cigar_vectors(oracle):
    # :946-987: 3 ops -> 12 bytes; all nine kinds round-trip; empty -> empty
    assert len(_encode([(10, "M"), (5, "I"), (3, "D")])) == 12
    all_kinds = [(k + 1, "MIDNSHP=X"[k]) for k in range(9)]
    raw = _encode(all_kinds)
    # the scan's binary_cigar column is exactly these bytes; the oracle formats the same packed words
    words = [v for (v,) in struct.iter_unpack("<I", raw)]
    assert words == _packed(all_kinds) and oracle.cigar_string(words) == "1M2I3D4N5S6H7P8=9X"
    assert _encode([]) == b""


# ---- tag_registry.rs:794-909 --------------------------------------------------------------------------------------
def test_known_tags_coverage(oracle):
    tags = oracle.known_tags()
    assert len(tags) == 63                                  # "Expected 63 standard SAM spec tags"
    for t in ("NM", "MD", "AS", "CB"):
        assert t in tags
    assert tags["NM"][0] == "i" and tags["NM"][1] == pa.int32()
    assert tags["MD"][0] == "Z" and tags["MD"][1] == pa.utf8()
    for t in ("NM", "AS", "CB"):
        assert tags[t][2]                                    # descriptions are present


def test_type_mapping(oracle):
    m = oracle.sam_tag_type_to_arrow_type
    assert m("i") == pa.int32() and m("I") == pa.uint32() and m("Z") == pa.utf8() and m("A") == pa.utf8()
    assert m("f") == pa.float32() and m("H") == pa.utf8() and m("B") == pa.list_(pa.int32())
    assert oracle.format_sam_tag_type("B", pa.list_(pa.uint16())) == "B:S"
    assert oracle.format_sam_tag_type("B", pa.list_(pa.uint8())) == "B:C"   # sam_array_subtype_from_arrow_type(List<UInt8>) == 'C'


def test_parse_tag_type_hints(oracle):
    m = oracle.parse_tag_type_hints(["pt:i", "de:f", "sv:Z", "ui:I", "ml:B:C", "cg:B:I"])
    assert len(m) == 6
    assert m["pt"] == ("i", pa.int32()) and m["de"] == ("f", pa.float32()) and m["sv"] == ("Z", pa.utf8())
    assert m["ui"] == ("I", pa.uint32()) and m["ml"] == ("B", pa.list_(pa.uint8())) and m["cg"] == ("B", pa.list_(pa.uint32()))


@pytest.mark.parametrize("bad", ["pt", "pt:X:extra", "pt:ii", "pt:X", "pt:z", "ml:B", "ml:B:Q"])
def test_parse_tag_type_hints_invalid(oracle, bad):
    with pytest.raises(Exception):
        oracle.parse_tag_type_hints([bad])


def test_infer_scalar_integer_types(oracle):
    # tag_registry.rs:891-908: c / C / s -> ('i', Int32); I -> ('I', UInt32)
    assert oracle.infer_type_from_value("c", -1) == ("i", pa.int32())
    assert oracle.infer_type_from_value("C", 1) == ("i", pa.int32())
    assert oracle.infer_type_from_value("s", -1) == ("i", pa.int32())
    assert oracle.infer_type_from_value("I", 2 ** 32 - 1) == ("I", pa.uint32())


def test_tag_write_kats(oracle):
    """bio-format-core/src/sam_tag_io.rs:1049-1089: a UInt32 column written under 'i' metadata is an Int32 aux value and u32::MAX
    does not fit; build_tag_data takes the SAM type from the field's bio.bam.tag.type."""
    import pyarrow as pa
    assert oracle.tag_aux_bytes("NM", "i", pa.uint32(), 42) == b"NMi" + (42).to_bytes(4, "little")
    with pytest.raises(oracle.TagWriteError, match="does not fit SAM type 'i'"):
        oracle.tag_aux_bytes("NM", "i", pa.uint32(), 2 ** 32 - 1)
    f = pa.field("NM", pa.int32(), True, metadata={"bio.bam.tag.tag": "NM", "bio.bam.tag.type": "i"})
    b = pa.RecordBatch.from_arrays([pa.array([7], pa.int32())], schema=pa.schema([f]))
    assert oracle.build_tag_data(b, 0) == b"NMi\x07\x00\x00\x00"
    # tag_registry.rs:883-900 parse_sam_tag_type
    assert oracle.parse_sam_tag_type("i") == ("i", None) and oracle.parse_sam_tag_type("B:C") == ("B", "C")
    for bad in ("ii", "B:CC", "B:x", "i:i", "B:C:S"):
        with pytest.raises(oracle.TagWriteError):
            oracle.parse_sam_tag_type(bad)
