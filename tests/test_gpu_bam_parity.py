"""GPU parity: HIP scan (through the C ABI) vs the CPU oracle on the reference's own BAM fixtures.
Bit-exact (Arrow logical equality: values, offsets, null positions) per partition and per batch."""
import os
import zlib

import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu

FIXTURES = [
    ("multi_chrom.bam", 421, None),
    ("multi_chrom_large.bam", 4277, None),
    ("10x_pbmc_tags.bam", 10, ["CB", "CR", "CY", "UB", "UR", "UY", "NH", "HI", "AS", "nM", "RE", "xf", "ts", "RG"]),
    ("bam_with_tags.bam", 14, ["NM", "MD", "MQ", "RG", "UQ", "XT", "XN", "OQ", "E2", "PG"]),
    ("nanopore_custom_tags.bam", 20, ["NM", "AS", "ns", "pa", "de", "tp", "cm", "s1", "ms", "nn", "rl"]),
    ("no_coor_only.bam", 2, ["CB", "CR"]),
]


def _cmp_batches(got, want, ctx):
    assert len(got) == len(want), (ctx, len(got), len(want))
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.num_rows == w.num_rows, (ctx, i, g.num_rows, w.num_rows)
        assert g.schema.names == w.schema.names, (ctx, g.schema.names, w.schema.names)
        for name in w.schema.names:
            gc, wc = g.column(name), w.column(name)
            assert gc.type == wc.type, (ctx, name, gc.type, wc.type)
            if not gc.equals(wc):
                gl, wl = gc.to_pylist(), wc.to_pylist()
                bad = [k for k in range(len(wl)) if gl[k] != wl[k]][:3]
                raise AssertionError((ctx, i, name, [(k, gl[k], wl[k]) for k in bad]))


def _assert_same_schema_and_metadata(got: pa.Schema, want: pa.Schema, ctx):
    """Field names, types, nullability, field metadata (bio.bam.tag.*) and schema metadata (bio.bam.*,
    bio.coordinate_system_zero_based); values that are JSON documents are compared parsed (key order inside an object is
    serde's / ours to choose, the content is not)."""
    import json
    assert got.equals(want, check_metadata=False), (ctx, got, want)

    def parsed(md):
        out = {}
        for k, v in (md or {}).items():
            v = v.decode()
            try:
                out[k.decode()] = json.loads(v) if v[:1] in "[{" else v
            except ValueError:
                out[k.decode()] = v
        return out
    assert parsed(got.metadata) == parsed(want.metadata), (ctx, parsed(got.metadata), parsed(want.metadata))
    for fg, fw in zip(got, want):
        assert parsed(fg.metadata) == parsed(fw.metadata), (ctx, fg.name, fg.metadata, fw.metadata)


@pytest.mark.parametrize("fname,count,tags", FIXTURES)
def test_inflate_matches_zlib(pkg, oracle, golden, fname, count, tags):
    data = open(os.path.join(golden, fname), "rb").read()
    want, _ = oracle.bgzf_inflate_all(data)
    got, ms = pkg.bgzf_inflate(data)
    assert got == want


@pytest.mark.parametrize("fname,count,tags", FIXTURES)
@pytest.mark.parametrize("zero_based", [True, False])
def test_sequential_scan(pkg, oracle, golden, fname, count, tags, zero_based):
    path = os.path.join(golden, fname)
    prov = pkg.BamTableProvider(path, None, zero_based, tags, index_path="")
    orc = oracle.BamOracle(path, zero_based=zero_based, tag_fields=tags, index_path=None)
    _assert_same_schema_and_metadata(prov.schema(), orc.schema, (fname, zero_based))
    for bs in (8192, 100):
        plan = prov.scan()
        assert plan.num_partitions() == 1
        got = list(plan.execute(0, bs))
        _, want = orc.execute_sequential(None, bs)
        assert sum(b.num_rows for b in got) == count
        _cmp_batches(got, want, (fname, zero_based, bs))


@pytest.mark.parametrize("fname,count,tags", FIXTURES)
@pytest.mark.parametrize("target", [1, 2, 3, 4, 8])
def test_indexed_partitions(pkg, oracle, golden, fname, count, tags, target):
    path = os.path.join(golden, fname)
    prov = pkg.BamTableProvider(path, None, True, tags)
    orc = oracle.BamOracle(path, zero_based=True, tag_fields=tags)
    plan = prov.scan(target_partitions=target)
    parts, residual = orc.scan(target_partitions=target)
    assert plan.num_partitions() == len(parts)
    total = 0
    for p in range(plan.num_partitions()):
        got = list(plan.execute(p, 64))
        _, want = orc.execute_partition(parts[p].regions, None, residual, 64)
        _cmp_batches(got, want, (fname, target, p, plan.partition_desc(p)))
        total += sum(b.num_rows for b in got)
    assert total == count  # bam/tests/indexed_read_test.rs:297-319


def test_projection_and_count_star(pkg, oracle, golden):
    path = os.path.join(golden, "multi_chrom.bam")
    prov = pkg.BamTableProvider(path)
    orc = oracle.BamOracle(path)
    plan = prov.scan(projection=[1, 2], target_partitions=2)
    assert plan.display() == "BamExec: projection=[chrom, start]"
    parts, _ = orc.scan(target_partitions=2)
    for p in range(plan.num_partitions()):
        got = list(plan.execute(p, 8192))
        _, want = orc.execute_partition(parts[p].regions, [1, 2], (), 8192)
        _cmp_batches(got, want, ("proj", p))
    plan = prov.scan(projection=[], target_partitions=3)
    rows = sum(b.num_rows for p in range(plan.num_partitions()) for b in plan.execute(p, 100))
    assert rows == 421


def test_region_filters(pkg, oracle, golden):
    path = os.path.join(golden, "multi_chrom_large.bam")
    prov = pkg.BamTableProvider(path)
    orc = oracle.BamOracle(path)
    cases = [
        [("chrom", "=", "chr1")],
        [("chrom", "=", "chr1"), ("start", ">=", 55000000), ("end", "<=", 55100000)],
        [("chrom", "in", ["chr2", "chrX"]), ("mapping_quality", ">=", 30)],
        [("chrom", "=", "chr2"), ("start", "between", (1, 200000000)), ("flags", "!=", 99)],
        [("mapping_quality", "between", (20, 60))],
        [("chrom", "=", "chr1"), ("start", ">", 10), ("start", "<", 5)],
        [("chrom", "in", ["chr1", "chr1", "chr2", "chr1"])],   # repeated regions: more selected rows than records
    ]
    for filters in cases:
        for target in (1, 3):
            plan = prov.scan(filters=filters, target_partitions=target)
            parts, residual = orc.scan(filters=filters, target_partitions=target)
            assert plan.num_partitions() == len(parts), filters
            for p in range(len(parts)):
                got = list(plan.execute(p, 1000))
                _, want = orc.execute_partition(parts[p].regions, None, residual, 1000)
                _cmp_batches(got, want, (filters, target, p))


@pytest.mark.parametrize("fname", ["multi_chrom_large.bam", "nanopore_custom_tags.bam"])
def test_binary_cigar_option(pkg, oracle, golden, fname):
    """bio.bam.binary_cigar=true: cigar is a Binary column of the raw little-endian ops (alignment_utils.rs:554-558)."""
    path = os.path.join(golden, fname)
    prov = pkg.BamTableProvider(path, None, True, None, binary_cigar=True)
    orc = oracle.BamOracle(path, zero_based=True, binary_cigar=True)
    _assert_same_schema_and_metadata(prov.schema(), orc.schema, (fname, "binary_cigar"))
    assert prov.schema().field("cigar").type == pa.binary()
    assert prov.schema().metadata[b"bio.bam.binary_cigar"] == b"true"
    for target in (1, 3):
        plan = prov.scan(target_partitions=target)
        parts, residual = orc.scan(target_partitions=target)
        assert plan.num_partitions() == len(parts)
        for p in range(len(parts)):
            got = list(plan.execute(p, 500))
            _, want = orc.execute_partition(parts[p].regions, None, residual, 500)
            _cmp_batches(got, want, (fname, "binary_cigar", target, p))


@pytest.mark.parametrize("fname,tags", [("10x_pbmc_tags.bam", ["CB", "xf", "ts", "NH", "ZZ"]), ("nanopore_custom_tags.bam", ["de", "tp", "rl", "NM"])])
def test_tag_type_inference_options(pkg, oracle, golden, fname, tags):
    """Schema and values with inference on (sample sizes 1 and 100), off, and with hints (table_provider.rs:72-126):
    inferred > hint > registry > Utf8."""
    path = os.path.join(golden, fname)
    for kw in (dict(infer_tag_types=True, infer_tag_sample_size=1), dict(infer_tag_types=True, infer_tag_sample_size=100),
               dict(infer_tag_types=False), dict(infer_tag_types=False, tag_type_hints=[f"{tags[1]}:Z", f"{tags[2]}:i"]),
               dict(infer_tag_types=True, infer_tag_sample_size=3, tag_type_hints=[f"{tags[0]}:Z"])):
        try:
            orc = oracle.BamOracle(path, zero_based=True, tag_fields=tags, index_path=None, **kw)
            want_err = None
        except Exception as e:  # noqa: BLE001
            orc, want_err = None, e
        if want_err is not None:
            with pytest.raises(Exception):
                pkg.BamTableProvider(path, None, True, tags, index_path="", **kw)
            continue
        prov = pkg.BamTableProvider(path, None, True, tags, index_path="", **kw)
        assert prov.schema().equals(orc.schema, check_metadata=True), (kw, prov.schema(), orc.schema)
        try:
            _, want = orc.execute_sequential(None, 8192)
        except Exception:  # a hint that contradicts the data: both sides refuse
            with pytest.raises(RuntimeError):
                list(prov.scan().execute(0, 8192))
            continue
        got = list(prov.scan().execute(0, 8192))
        _cmp_batches(got, want, (fname, kw))


def test_differential_fuzz_of_indexed_scans(pkg, oracle):
    """tools/fuzz_bam_indexed.py, a fixed number of files of two seeds: random coordinate-sorted BAMs (crowded and empty bins,
    reference spans across bin and linear-index boundaries, placed-unmapped and unplaced reads, members of 200 .. 60 000 bytes)
    with a BAI built by the tool, scanned with random target_partitions, filters, projections, batch sizes, coordinate
    systems and pipeline chunk sizes -- partition plans and every partition's batches against the oracle; a scan one side
    refuses (a region on a reference the file does not have) the other refuses too."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_bam_indexed as F
    tot = dict(files=0, scans=0, partitions=0, rows=0, refused_by_both=0)
    for seed in (11, 12):
        t, failures = F.run(pkg, seed=seed, max_files=50, verbose=False)
        assert not failures, failures[:3]
        for k in tot:
            tot[k] += t[k]
    assert tot["scans"] > 150 and tot["rows"] > 5000, tot
    from conftest import report_size
    report_size("test_differential_fuzz_of_indexed_scans", **tot)
