"""CPU: pin the oracle against everything the reference's own tests hold for this path
(SURVEY 8c): record counts, per-chromosome counts, partition-count invariance, tag schema shapes,
CIGAR known answers, no-coor handling.  Also C oracle == Python oracle."""
import os

import pyarrow as pa
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _counts(o, path, target, **kw):
    b = o.BamOracle(path, **kw)
    parts, res = b.scan(target_partitions=target)
    per = {}
    n = 0
    for p in parts:
        _, bs = b.execute_partition(p.regions, [1], res, 8192)
        for x in bs:
            n += x.num_rows
            for c in x.column(0).to_pylist():
                per[c] = per.get(c, 0) + 1
    return n, per, len(parts)


def test_multi_chrom_counts(oracle):
    # bam/tests/indexed_read_test.rs:76,108,121
    n, per, _ = _counts(oracle, os.path.join(G, "multi_chrom.bam"), 1)
    assert n == 421 and per == {"chr1": 160, "chr2": 159, "chrX": 102}


@pytest.mark.parametrize("target", [1, 2, 3, 4, 8])
def test_partition_count_invariance(oracle, target):
    # bam/tests/indexed_read_test.rs:297-319, indexed_read_large_test.rs
    assert _counts(oracle, os.path.join(G, "multi_chrom.bam"), target)[0] == 421
    n, per, _ = _counts(oracle, os.path.join(G, "multi_chrom_large.bam"), target)
    assert n == 4277 and per == {"chr1": 1662, "chr2": 1694, "chrX": 921}


def test_indexed_equals_sequential(oracle):
    # bam/tests/indexed_read_test.rs:208-237
    p = os.path.join(G, "multi_chrom.bam")
    b = oracle.BamOracle(p)
    _, seq = b.execute_sequential([0, 1, 2], 1 << 20)
    rows = sorted(zip(*[seq[0].column(i).to_pylist() for i in range(3)]))
    parts, res = b.scan(target_partitions=4)
    got = []
    for part in parts:
        _, bs = b.execute_partition(part.regions, [0, 1, 2], res, 1 << 20)
        for x in bs:
            got += list(zip(*[x.column(i).to_pylist() for i in range(3)]))
    assert sorted(got) == rows


def test_no_coor_partition(oracle):
    # bam/tests/indexed_read_test.rs:244-271
    b = oracle.BamOracle(os.path.join(G, "no_coor_only.bam"), tag_fields=["CB", "CR"])
    parts, res = b.scan(target_partitions=4)
    rows = []
    for p in parts:
        _, bs = b.execute_partition(p.regions, None, res, 8192)
        for x in bs:
            rows += x.to_pylist()
    assert len(rows) == 2
    assert all(r["chrom"] is None for r in rows)
    assert all(r["CB"] is not None and r["CR"] is not None for r in rows)


def test_tag_schema_shapes(oracle):
    # bam/tests/tag_tests.rs:30,59,195-209,243,311-342,495-511
    p = os.path.join(G, "bam_with_tags.bam")
    assert len(oracle.BamOracle(p).schema) == 12
    s = oracle.BamOracle(p, tag_fields=["NM", "MD"]).schema
    assert len(s) == 14 and s.field("NM").type == pa.int32() and s.field("NM").nullable
    s = oracle.BamOracle(p, tag_fields=["UNKNOWN_TAG"], infer_tag_types=False).schema
    f = s.field("UNKNOWN_TAG")
    assert len(s) == 13 and f.type == pa.utf8() and f.nullable
    assert f.metadata[b"bio.bam.tag.tag"] == b"UNKNOWN_TAG" and f.metadata[b"bio.bam.tag.type"] == b"Z"
    assert len(oracle.BamOracle(p, tag_fields=["NM", "MD", "AS", "RG"]).schema) == 16
    tenx = ["CB", "CR", "CY", "UB", "UR", "UY", "NH", "HI", "AS", "nM", "RE", "xf", "ts", "RG"]
    b = oracle.BamOracle(os.path.join(G, "10x_pbmc_tags.bam"), tag_fields=tenx)
    assert len(b.schema) == 26
    for t in ("NH", "HI", "AS", "nM", "ts", "xf"):
        assert b.schema.field(t).type == pa.int32(), t
    for t in ("RG", "CR", "CY", "CB", "UR", "UY", "UB", "RE"):
        assert b.schema.field(t).type == pa.utf8(), t
    _, bs = b.execute_sequential()
    rows = bs[0].to_pylist()
    assert len(rows) == 10
    assert all(r["RG"].startswith("10k_") for r in rows)          # tag_tests.rs:372
    assert all(len(r["RE"]) == 1 for r in rows if r["RE"] is not None)
    assert any(r["ts"] is None for r in rows) and any(r["ts"] is not None for r in rows)
    alltags = ["NM", "MD", "MQ", "XT", "RG", "PG", "UQ", "OQ", "E2", "OC", "OP", "XN", "ZQ"]
    b = oracle.BamOracle(p, tag_fields=alltags)
    assert len(b.schema) == 25
    for t, ty in (("NM", pa.int32()), ("MD", pa.utf8()), ("XT", pa.int32()), ("MQ", pa.int32()), ("RG", pa.utf8()), ("OP", pa.int32())):
        assert b.schema.field(t).type == ty, t
    _, bs = b.execute_sequential()
    assert sum(x.num_rows for x in bs) == 14


def test_cigar_known_answers(oracle):
    # bio-format-core/src/alignment_utils.rs:818-870
    enc = lambda ops: [(n << 4) | "MIDNSHP=X".index(c) for n, c in ops]
    assert oracle.cigar_string(enc([(10, "M"), (5, "I"), (3, "D")])) == "10M5I3D"
    assert oracle.cigar_string(enc([(150, "M")])) == "150M"
    assert oracle.cigar_string([]) == ""
    assert oracle.cigar_string(enc([(7, "S"), (100, "M"), (3, "S")])) == "7S100M3S"
    assert oracle.cigar_string(enc([(100, "N"), (2, "H"), (1, "P"), (4, "="), (5, "X")])) == "100N2H1P4=5X"
    assert oracle.ref_span(enc([(10, "M"), (5, "I"), (3, "D"), (7, "S"), (4, "N"), (2, "="), (1, "X")])) == 20


def test_coordinate_systems(oracle):
    # start shifts with the coordinate system, end never does (genomic_filter.rs:240-242)
    p = os.path.join(G, "multi_chrom.bam")
    _, z = oracle.BamOracle(p, zero_based=True).execute_sequential([2, 3, 8])
    _, o1 = oracle.BamOracle(p, zero_based=False).execute_sequential([2, 3, 8])
    zs, os_ = z[0].column(0).to_pylist(), o1[0].column(0).to_pylist()
    assert all(a + 1 == b for a, b in zip(zs, os_) if a is not None)
    assert z[0].column(1).equals(o1[0].column(1))


def test_genomic_region_extraction(oracle):
    # bio-format-core/src/genomic_filter.rs:376-390 (start >= X AND end <= Y, zero-based -> [X+1, Y])
    r, unsat = oracle.extract_genomic_regions([("chrom", "=", "chr1"), ("start", ">=", 999), ("end", "<=", 2000)], True)
    assert not unsat and [(x.chrom, x.start, x.end) for x in r] == [("chr1", 1000, 2000)]
    r, unsat = oracle.extract_genomic_regions([("chrom", "in", ["chr2", "chr1"])], True)
    assert [x.chrom for x in r] == ["chr1", "chr2"]
    r, unsat = oracle.extract_genomic_regions([("chrom", "=", "chr1"), ("start", ">", 100), ("start", "<", 50)], False)
    assert unsat and r == []


def test_c_oracle_equals_python_oracle(oracle):
    import sys
    sys.path.insert(0, os.path.dirname(oracle.__file__))
    import c_oracle
    for f in ("multi_chrom.bam", "nanopore_custom_tags.bam", "no_coor_only.bam", "bam_with_tags.bam", "10x_pbmc_tags.bam"):
        for zb in (True, False):
            st, cols = c_oracle.scan(open(os.path.join(G, f), "rb").read(), zb, 3)
            b = oracle.BamOracle(os.path.join(G, f), zero_based=zb, index_path=None)
            _, bs = b.execute_sequential(None, 1 << 30)
            t = pa.Table.from_batches(bs)
            for name, got in cols.items():
                if pa.types.is_large_string(got.type):
                    got = got.cast(pa.utf8())
                assert got.equals(t.column(name).combine_chunks()), (f, name)


def _col_digest(arr):
    """(value bytes, sum of value bytes, non-NULL rows) of one column, the order-independent digest the streaming C
    baseline folds per batch."""
    arr = arr.combine_chunks() if isinstance(arr, pa.ChunkedArray) else arr
    n_valid = len(arr) - arr.null_count
    if pa.types.is_string(arr.type) or pa.types.is_large_string(arr.type):
        off_t = "q" if pa.types.is_large_string(arr.type) else "i"
        import struct
        offs = arr.buffers()[1]
        w = 8 if off_t == "q" else 4
        lo = struct.unpack_from("<" + off_t, offs, arr.offset * w)[0]
        hi = struct.unpack_from("<" + off_t, offs, (arr.offset + len(arr)) * w)[0]
        data = arr.buffers()[2].to_pybytes()[lo:hi] if arr.buffers()[2] is not None else b""
        return len(data), sum(data), n_valid
    data = arr.buffers()[1].to_pybytes()[arr.offset * 4:(arr.offset + len(arr)) * 4]
    # NULL slots hold 0 in both implementations
    return len(data), sum(data), n_valid


@pytest.mark.parametrize("threads", [1, 2, 3, 7])
def test_streaming_c_baseline_equals_materialising_oracle(oracle, threads):
    """oracle_bam_scan_stream (bench.py's cpu_baseline: one thread per partition, member by member, per-batch builders --
    the reference's executor shape) returns the rows the materialising C oracle and the Python oracle return: same row
    count, and per column the same value bytes, byte sum and NULL count, for any partition count and batch size."""
    import sys
    sys.path.insert(0, os.path.dirname(oracle.__file__))
    import c_oracle
    for f in ("multi_chrom.bam", "multi_chrom_large.bam", "nanopore_custom_tags.bam", "no_coor_only.bam", "10x_pbmc_tags.bam"):
        data = open(os.path.join(G, f), "rb").read()
        for zb, bs in ((True, 8192), (False, 100), (True, 1)):
            _, cols = c_oracle.scan(data, zb, 2)
            plan = c_oracle.stream_plan(data, threads)
            r = c_oracle.stream_scan(data, plan, zb, bs)
            n = len(cols["name"])
            assert r["n_rows"] == n, (f, threads, r["n_rows"], n)
            assert r["n_batches"] >= (n + bs - 1) // bs and r["n_batches"] <= (n + bs - 1) // bs + threads
            for name, arr in cols.items():
                nb, sm, nv = _col_digest(arr)
                assert (r["col_bytes"][name], r["col_sum"][name], r["n_valid"][name]) == (nb, sm, nv), (f, threads, zb, bs, name)
