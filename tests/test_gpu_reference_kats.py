"""Device side of the reference's known-answer tests (tests/test_cpu_reference_kats.py holds the vectors and the host
side): the residual filter kernel (k_row_flags), bioscan_supports_filters_pushdown, the tag back-fill of
load_record_tags, the binary CIGAR column, IN lists of any length."""
import os
import struct

import pytest

import bam_build as bb
from test_gpu_bam_parity import _cmp_batches

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _rows(plan):
    out = []
    for p in range(plan.num_partitions()):
        for b in plan.execute(p, 8192):
            out += list(zip(b.column("name").to_pylist(), b.column("chrom").to_pylist(), b.column("start").to_pylist(),
                            b.column("mapping_quality").to_pylist()))
    return out


def test_record_filter_kats_on_device(pkg, oracle):
    """record_filter.rs:358-520 with a real record in the role of TestRecord: the filters of each reference test, built
    from that record's own values, keep or drop it exactly as the reference asserts -- and the whole scan equals the oracle's."""
    path = os.path.join(G, "multi_chrom_large.bam")
    prov = pkg.BamTableProvider(path)
    orc = oracle.BamOracle(path)
    base = _rows(prov.scan(filters=[("chrom", "=", "chr1")], target_partitions=1))
    name, chrom, start, mapq = next(r for r in base if r[3] not in (0, 255))
    other = "chr2"
    cases = [
        ([("chrom", "=", chrom)], True),                                         # test_evaluate_chrom_eq
        ([("chrom", "=", chrom), ("chrom", "!=", chrom)], False),                # (... a miss on the same index path)
        ([("chrom", "=", chrom), ("mapping_quality", ">=", mapq)], True),        # test_evaluate_numeric_gte
        ([("chrom", "=", chrom), ("mapping_quality", ">=", mapq + 1)], False),
        ([("chrom", "in", [chrom, other])], True),                               # test_in_list_filter
        ([("chrom", "=", chrom), ("mapping_quality", "in", [mapq + 1, mapq + 2])], False),
        ([("chrom", "=", chrom), ("mapping_quality", "not in", [mapq + 1, None])], False),   # NOT IN with NULL: UNKNOWN
        ([("chrom", "=", chrom), ("mapping_quality", "not in", [mapq + 1])], True),
        ([("chrom", "=", chrom), ("mapping_quality", "=", None)], False),        # NULL literal never passes
        ([("chrom", "=", chrom), ("start", "between", (start, start))], True),
        ([("chrom", "=", chrom), ("flags", "not between", (0, 65535))], False),
    ]
    for filters, present in cases:
        for target in (1, 3):
            plan = prov.scan(filters=filters, target_partitions=target)
            got = _rows(plan)
            assert ((name, chrom, start, mapq) in got) is present, (filters, target)
            parts, residual = orc.scan(filters=filters, target_partitions=target)
            assert plan.num_partitions() == len(parts)
            for p in range(plan.num_partitions()):
                _, want = orc.execute_partition(parts[p].regions, None, residual, 8192)
                _cmp_batches(list(plan.execute(p, 8192)), want, (filters, target, p))


def test_in_lists_longer_than_eight_literals(pkg, oracle):
    """The device filter takes lists of any length (eight literals per term, continued terms): same rows as the oracle."""
    path = os.path.join(G, "multi_chrom_large.bam")
    prov = pkg.BamTableProvider(path)
    orc = oracle.BamOracle(path)
    mapqs = list(range(0, 61, 3))            # 21 literals
    chroms = ["chr1", "chrX", "chr2"] * 4     # 12 literals (a name the header does not hold is an error, as in the reference)
    for filters in ([("chrom", "in", chroms)], [("chrom", "=", "chr1"), ("mapping_quality", "in", mapqs)],
                    [("chrom", "=", "chr1"), ("mapping_quality", "not in", mapqs)],
                    [("chrom", "=", "chr1"), ("mapping_quality", "not in", mapqs + [None])],
                    [("chrom", "in", chroms), ("flags", "in", list(range(60, 200)))]):
        plan = prov.scan(filters=filters, target_partitions=2)
        parts, residual = orc.scan(filters=filters, target_partitions=2)
        assert plan.num_partitions() == len(parts), filters
        n = 0
        for p in range(plan.num_partitions()):
            got = list(plan.execute(p, 512))
            _, want = orc.execute_partition(parts[p].regions, None, residual, 512)
            _cmp_batches(got, want, (filters, p))
            n += sum(b.num_rows for b in got)
        if filters[-1][1] == "not in" and None in filters[-1][2]:
            assert n == 0


def test_can_push_down_on_device_provider(pkg):
    # record_filter.rs:455-471 + table_provider.rs:941-962
    prov = pkg.BamTableProvider(os.path.join(G, "multi_chrom.bam"))
    assert prov.supports_filters_pushdown([("chrom", "=", "chr1"), ("start", ">=", 1000), ("mapping_quality", ">=", 30)]) == ["Inexact"] * 3
    assert prov.supports_filters_pushdown([("sequence", "<", "A"), ("nope", "=", 1)]) == ["Unsupported", "Unsupported"]
    noidx = pkg.BamTableProvider(os.path.join(G, "multi_chrom.bam"), index_path="")
    assert noidx.supports_filters_pushdown([("chrom", "=", "chr1")]) == ["Inexact"]   # record-level, not index-level


def test_load_record_tags_backfills_nulls(pkg, oracle, tmp_path):
    """sam_tag_io.rs:1094-1128: a record with NM:i:3 and no MD -> NM = 3, MD NULL (and the reverse on the next record)."""
    refs = [("chr1", 1000)]
    recs = [bb.record(name="a", aux_bytes=bb.aux("NM", "i", 3)),
            bb.record(name="b", aux_bytes=bb.aux("MD", "Z", "10")),
            bb.record(name="c")]
    path = str(tmp_path / "tags.bam")
    open(path, "wb").write(bb.bam(refs, recs))
    prov = pkg.BamTableProvider(path, None, True, ["NM", "MD"], index_path="")
    got = list(prov.scan().execute(0, 8192))
    assert got[0].column("NM").to_pylist() == [3, None, None]
    assert got[0].column("MD").to_pylist() == [None, "10", None]
    orc = oracle.BamOracle(path, zero_based=True, tag_fields=["NM", "MD"], index_path=None)
    _cmp_batches(got, orc.execute_sequential(None, 8192)[1], "backfill")


def test_binary_cigar_column_is_the_encoded_ops(pkg, tmp_path):
    """alignment_utils.rs:946-987: encode_cigar_ops_to_binary writes len << 4 | op as LE u32s; the binary_cigar column of a
    scan is exactly those bytes for all nine op kinds, empty for an empty CIGAR."""
    all_kinds = tuple((k + 1, "MIDNSHP=X"[k]) for k in range(9))
    recs = [bb.record(name="all", cigar=all_kinds, seq="A" * 30), bb.record(name="none", cigar=(), seq="ACGT"),
            bb.record(name="three", cigar=((10, "M"), (5, "I"), (3, "D")), seq="A" * 15)]
    path = str(tmp_path / "cig.bam")
    open(path, "wb").write(bb.bam([("chr1", 100000)], recs))
    prov = pkg.BamTableProvider(path, None, True, None, True, index_path="")   # binary_cigar = True
    got = list(prov.scan().execute(0, 8192))[0].column("cigar").to_pylist()
    enc = lambda ops: b"".join(struct.pack("<I", (n << 4) | "MIDNSHP=X".index(k)) for n, k in ops)  # noqa: E731
    assert got == [enc(all_kinds), b"", enc(((10, "M"), (5, "I"), (3, "D")))]
    assert len(got[2]) == 12
    text = pkg.BamTableProvider(path, None, True, None, False, index_path="")
    assert list(text.scan().execute(0, 8192))[0].column("cigar").to_pylist() == ["1M2I3D4N5S6H7P8=9X", "", "10M5I3D"]


@pytest.mark.parametrize("hints,ok", [(["pt:i", "de:f", "sv:Z", "ui:I", "ml:B:C", "cg:B:I"], True), (["pt"], False), (["pt:X:extra"], False),
                                      (["pt:ii"], False), (["pt:X"], False), (["pt:z"], False), (["ml:B"], False), (["ml:B:Q"], False)])
def test_tag_type_hints_on_provider(pkg, hints, ok):
    # tag_registry.rs:848-877 through BamTableProvider::new (table_provider.rs:395-399: a bad hint is a configuration error)
    path = os.path.join(G, "nanopore_custom_tags.bam")
    tags = [h.split(":")[0] for h in hints]
    if ok:
        prov = pkg.BamTableProvider(path, None, True, tags, False, False, 100, hints)
        types = {f.name: str(f.type) for f in prov.schema()}
        assert types["pt"] == "int32" and types["de"] == "float" and types["sv"] == "string" and types["ui"] == "uint32"
        assert types["ml"] == "list<item: uint8>" and types["cg"] == "list<item: uint32>"
    else:
        with pytest.raises(pkg.BioscanError):
            pkg.BamTableProvider(path, None, True, tags, False, False, 100, hints)
