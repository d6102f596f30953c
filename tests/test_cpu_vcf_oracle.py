"""Pins oracle/vcf_oracle.py against the values the reference's own VCF tests assert (CPU only)."""
import os
import sys

import pyarrow as pa
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
sys.path.insert(0, HERE)
import vcf_oracle as V  # noqa: E402
import vcf_cases as C  # noqa: E402

GOLD = os.path.join(HERE, "golden")


class OracleTable:
    def __init__(self, path, **kw):
        self.o = V.VcfOracle(path, **kw)

    def column_names(self):
        return self.o.schema.names

    def read(self, names=None, filters=(), target_partitions=1, limit=None, batch_size=8192):
        proj = None if names is None else [self.o.schema.get_field_index(n) for n in names]
        plan = self.o.scan(projection=proj, filters=filters, limit=limit, target_partitions=target_partitions)
        cols = {n: [] for n in (names if names is not None else self.o.schema.names)}
        self.rows = 0
        for p in range(self.o.num_partitions(plan)):
            _, bs = self.o.execute(plan, p, batch_size)
            for b in bs:
                self.rows += b.num_rows
                for n in cols:
                    cols[n].extend(b.column(b.schema.get_field_index(n)).to_pylist())
        return cols


def test_reference_kats(tmp_path):
    k = [0]

    def make(text, **kw):
        k[0] += 1
        p = tmp_path / f"case{k[0]}.vcf"
        p.write_text(text)
        return OracleTable(str(p), **kw)
    C.check_reference_kats(make)


def test_indexed_counts():
    # indexed_read_test.rs:99-145, indexed_read_large_test.rs:50-95
    for name, per in (("multi_chrom.vcf.gz", 500), ("multi_chrom_large.vcf.gz", 5000)):
        t = OracleTable(os.path.join(GOLD, name))
        for tp in (1, 2, 3, 4, 8):
            t.read(["chrom"], target_partitions=tp)
            assert t.rows == 2 * per
            r = t.read(["chrom"], filters=[("chrom", "=", "21")], target_partitions=tp)
            assert t.rows == per and set(r["chrom"]) == {"21"}
            t.read(["chrom"], filters=[("chrom", "in", ["21", "22"])], target_partitions=tp)
            assert t.rows == 2 * per
        r = t.read(["chrom"], filters=[("chrom", "=", "21"), ("start", ">=", 5009999), ("start", "<=", 5029999)], target_partitions=4)
        assert 0 < t.rows < per
    # indexed_read_test.rs:222-257 (1-based provider, info_fields=[])
    t = OracleTable(os.path.join(GOLD, "multi_chrom.vcf.gz"), info_fields=[], zero_based=False)
    t.read([], filters=[("chrom", "=", "21"), ("start", "=", 5000100)], target_partitions=4)
    assert t.rows == 1
    plan = t.o.scan(filters=[("chrom", "=", "21"), ("start", "=", 5000100), ("start", ">", 5000100)])
    assert plan["kind"] == "empty"


def test_indexed_metadata_and_limits():
    o = V.VcfOracle(os.path.join(GOLD, "multi_chrom.vcf.gz"))
    import json
    assert json.loads(o.schema.metadata[b"bio.vcf.contigs.indexed"]) == ["21", "22"]  # indexed_read_test.rs:319-356
    t = OracleTable(os.path.join(GOLD, "multi_chrom.vcf.gz"))
    t.read(["chrom"], filters=[("chrom", "=", "21")], limit=5)  # limit_and_indexed_projection_test.rs:233-238
    assert t.rows == 5
    t.read(["chrom", "start"], filters=[("chrom", "=", "22")], limit=1)
    assert t.rows == 1
    t.read(["chrom"], filters=[("chrom", "=", "21")], limit=9999)
    assert t.rows == 500
    assert o.scan(limit=0)["kind"] == "empty"


def test_tbi_estimates():
    # storage.rs:1064-1154 unit tests
    o = V.VcfOracle(os.path.join(GOLD, "multi_chrom.vcf.gz"))
    names = o.tbi.names
    regions = [V.GenomicRegion(n) for n in names]
    est = V.estimate_sizes_from_tbi(o.tbi, regions, names, [])
    assert all(e.contig_length is not None and e.contig_length > 10_000 for e in est)
    lens = [999 - i for i in range(len(names))]
    est2 = V.estimate_sizes_from_tbi(o.tbi, regions, names, lens)
    assert [e.contig_length for e in est2] == lens
    mis = ["__extra_before_1", "__extra_before_2"] + names
    est3 = V.estimate_sizes_from_tbi(o.tbi, regions, mis, [0] * len(mis))
    assert [e.estimated_bytes for e in est3] == [e.estimated_bytes for e in est]
    assert [e.nonempty_bin_positions for e in est3] == [e.nonempty_bin_positions for e in est]
    # both contigs share the single data block: 21 -> 0, 22 -> 7163
    assert [e.estimated_bytes for e in est] == [0, 7163]


def test_real_multisample_fixture():
    # format_columns_test.rs:378-398: AD declared Number=. ; 6 rows
    t = OracleTable(os.path.join(GOLD, "head_106667_tail_6.vcf"), info_fields=[], format_fields=["GT", "AD", "DP", "GQ", "PL"])
    r = t.read(["chrom", "start", "genotypes"])
    assert t.rows == 6
    assert len(r["genotypes"][0]["GT"]) == len(t.o.source_samples)


def test_udf_kats():
    # udfs.rs:995-1162: the test batch and the values its unit tests assert
    L = pa.list_(pa.field("item", pa.int32(), True))
    gq = pa.array([[30, 20, 10], [5, None, 15]], type=L)
    dp = pa.array([[50, 30, 20], [10, 200, 100]], type=L)
    assert V.list_avg(gq).to_pylist() == [20.0, 10.0]
    assert V.list_gte(gq, 15).to_pylist() == [[True, True, False], [False, None, True]]
    assert V.list_and(V.list_gte(gq, 10), V.list_lte(dp, 100)).to_pylist() == [[True, True, True], [False, False, True]]
    # NULL list -> NULL; empty / all-null list -> NULL average (udfs.rs:73-86)
    a = pa.array([None, [], [None]], type=L)
    assert V.list_avg(a).to_pylist() == [None, None, None]
    assert V.list_gte(a, 1).to_pylist() == [None, [], [None]]
    F = pa.list_(pa.field("item", pa.float32(), True))
    assert V.list_avg(pa.array([[1.5, 2.5], [None]], type=F)).to_pylist() == [2.0, None]


def test_choose_effective_batch_size():
    # physical_exec.rs:81-137 (1000 samples x 3 fields -> 33)
    assert V.choose_effective_batch_size(8192, True, 3, 1000, 1000) == 33
    assert V.choose_effective_batch_size(8192, False, 3, 1000, 1000) == 8192
    assert V.choose_effective_batch_size(8192, True, 3, 1, 1) == 8192
    assert V.choose_effective_batch_size(8192, True, 2, 2, 2) == 8192
