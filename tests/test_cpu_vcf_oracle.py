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
    # vcf_set_gts (udfs.rs:1112-1139): mask false -> "./.", NULL mask element keeps the GT
    G = pa.list_(pa.field("item", pa.utf8(), True))
    gt = pa.array([["0/1", "1/1", "0/0"], ["./.", "0/1", "1/1"]], type=G)
    assert V.vcf_set_gts(gt, V.list_gte(gq, 15), "./.").to_pylist() == [["0/1", "1/1", "./."], ["./.", "0/1", "1/1"]]
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


def _synth(tmp_path, *args):
    import json
    import subprocess
    root = os.path.join(HERE, "..")
    exe = os.path.join(root, "tools", "_build", "synth_vcf")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "tools")], stdout=subprocess.DEVNULL)
    return json.loads(subprocess.check_output([exe] + [str(a) for a in args]))


def _kinds(o):
    out = []
    for name in o.info_fields:
        t = o.schema.field(name).type
        if pa.types.is_list(t):
            et = t.value_type
            out.append((name, "list_int" if et == pa.int32() else "list_float" if et == pa.float32() else "list_string"))
        else:
            out.append((name, "int" if t == pa.int32() else "float" if t == pa.float32() else "flag" if t == pa.bool_() else "string"))
    return out


def test_c_vcf_oracle_matches_python_oracle(tmp_path):
    """oracle/bioscan_oracle.c: oracle_vcf_scan_mem (the CPU baseline of the VCF bench lines) reproduces the column
    checksums of the Python oracle on synthetic config-3 and config-4 style files."""
    import c_oracle
    import numpy as np
    for mode, args, ns in (("sites", (3000, 3), 0), ("samples", (120, 40, 3), 40)):
        path = str(tmp_path / f"{mode}.vcf.gz")
        _synth(tmp_path, mode, path, *args)
        o = V.VcfOracle(path, index_path=None)
        _, bs = o.execute(o.scan(), 0)
        t = pa.Table.from_batches(bs)
        data = open(path, "rb").read()
        for threads in (1, 3):
            r = c_oracle.vcf_scan(data, _kinds(o), ns, True, threads, x0=o.data_start)
            assert r["n_rows"] == t.num_rows
            assert r["sum_start"] == sum(t.column("start").to_pylist()) and r["sum_end"] == sum(t.column("end").to_pylist())
            assert r["core_str_bytes"] == sum(len(s) for c in ("chrom", "id", "ref", "alt", "filter") for s in t.column(c).to_pylist())
            q = [x for x in t.column("qual").to_pylist() if x is not None]
            assert r["n_qual_valid"] == len(q) and abs(r["sum_qual"] - sum(q)) < 1e-6 * max(1.0, sum(q))
            ints = flt = strs = flags = elems = 0
            for name, kind in _kinds(o):
                for v in t.column(name).to_pylist():
                    if v is None:
                        continue
                    if kind == "int":
                        ints += v
                    elif kind == "flag":
                        flags += bool(v)
                    elif kind == "string":
                        strs += len(v)
                    elif kind == "float":
                        flt += 1
                    else:
                        elems += len(v)
                        for x in v:
                            if x is None:
                                continue
                            if kind == "list_int":
                                ints += x
                            elif kind == "list_float":
                                flt += 1
                            else:
                                strs += len(x)
            assert (r["info_int_sum"], r["info_float_valid"], r["info_str_bytes"], r["info_flag_true"], r["info_list_elems"]) == \
                (ints, flt, strs, flags, elems)
            if ns:
                g = t.column("genotypes").combine_chunks()
                gq, dp, gt = g.field("GQ"), g.field("DP"), g.field("GT")
                fq = [x for row in gq.to_pylist() for x in row]
                fd = [x for row in dp.to_pylist() for x in row]
                fg = [x for row in gt.to_pylist() for x in row]
                assert r["cells"] == len(fq)
                assert (r["gq_sum"], r["gq_valid"]) == (sum(x for x in fq if x is not None), sum(x is not None for x in fq))
                assert (r["dp_sum"], r["dp_valid"]) == (sum(x for x in fd if x is not None), sum(x is not None for x in fd))
                assert (r["gt_bytes"], r["gt_valid"]) == (sum(len(x) for x in fg if x is not None), sum(x is not None for x in fg))
                a = V.list_avg(gq).to_pylist()
                assert r["avg_gq_valid"] == sum(x is not None for x in a)
                assert abs(r["avg_gq_sum"] - sum(x for x in a if x is not None)) < 1e-9 * max(1.0, r["avg_gq_sum"])
                assert r["gq_gte_true"] == sum(bool(x) for row in V.list_gte(gq, 10).to_pylist() for x in row)
                assert r["dp_gte_true"] == sum(bool(x) for row in V.list_gte(dp, 10).to_pylist() for x in row)
                assert r["dp_lte_true"] == sum(bool(x) for row in V.list_lte(dp, 200).to_pylist() for x in row)


def test_csi_index_contributes_names_only():
    """indexed_read_test.rs:380-417: a `.csi` companion populates `bio.vcf.contigs.indexed`; the text-VCF reader then
    hands it to the tabix reader (storage.rs:766), so an indexed partition fails when it is executed."""
    import json
    o = V.VcfOracle(os.path.join(GOLD, "multi_chrom_csi.vcf.gz"))
    assert o.index_path.endswith(".csi")
    assert sorted(json.loads(o.schema.metadata[b"bio.vcf.contigs.indexed"])) == ["21", "22"]
    plan = o.scan(target_partitions=4)
    assert plan["kind"] == "indexed" and o.num_partitions(plan) == 2
    assert [a.total_estimated_bytes for a in plan["assignments"]] == [1, 1]
    with pytest.raises(V.VcfError):
        o.execute(plan, 0)
    # without the index the same file scans sequentially: 1000 variants like multi_chrom.vcf.gz
    o2 = V.VcfOracle(os.path.join(GOLD, "multi_chrom_csi.vcf.gz"), index_path=None)
    assert b"bio.vcf.contigs.indexed" not in o2.schema.metadata
    _, batches = o2.execute(o2.scan())
    assert sum(b.num_rows for b in batches) == 1000


def test_allele_stat_kats():
    """vcf_an / vcf_ac / vcf_af of the oracle against the reference's unit tests (udfs.rs:1165-1562)."""
    import allele_stat_cases as K
    for s, want in K.PARSE_KATS:
        assert V.parse_gt_alleles(s) == want, s
    for s, want in K.ALT_KATS:
        assert V.count_alt_alleles(s) == want, s
    for name, rows, alt, an, ac, af in K.STAT_KATS:
        gt, alt_arr = K.arrays(rows, alt)
        assert V.vcf_an(gt).to_pylist() == an, name
        assert V.vcf_ac(gt, alt_arr).to_pylist() == ac, name
        got = V.vcf_af(gt, alt_arr).to_pylist()
        assert len(got) == len(af), name
        for g, w in zip(got, af):
            assert len(g) == len(w) and all((a is None and b is None) or abs(a - b) < 0.001 for a, b in zip(g, w)), (name, g, w)


def test_lazy_columns_follow_the_reference(tmp_path):
    """noodles' record is lazy and the reference asks it per projected column (physical_exec.rs:758-823): a QUAL that does not
    parse, an INFO entry that does not parse, a genotype that does not parse are errors only for a scan that asks for that
    column -- INFO as a whole (every entry is typed once any INFO column is asked for and at least one INFO field is selected,
    :552-571), a FORMAT value as a whole per selected sample (:1661-1666), a genotype only when GT is a selected field
    (:1668-1676); a key that occurs twice breaks only the batch that holds its column (:572-575)."""
    hdr = ("##fileformat=VCFv4.3\n##contig=<ID=c,length=1000>\n"
           '##INFO=<ID=DP,Number=1,Type=Integer,Description="d">\n##INFO=<ID=MQ,Number=1,Type=Float,Description="m">\n'
           '##INFO=<ID=END,Number=1,Type=Integer,Description="e">\n'
           '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">\n##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="q">\n'
           "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS0\n")

    def scan(body, names, **kw):
        p = tmp_path / "t.vcf"
        p.write_text(hdr + body)
        o = V.VcfOracle(str(p), **kw)
        proj = None if names is None else [o.schema.get_field_index(n) for n in names]
        plan = o.scan(projection=proj, filters=[], limit=None, target_partitions=1)
        return o.execute(plan, 0, 100)[1]

    bad_qual = "c\t5\t.\tA\tT\tabc\t.\tDP=1\tGT:GQ\t0/1:5\n"
    assert scan(bad_qual, ["chrom", "start", "DP"])[0].num_rows == 1
    with pytest.raises(ValueError):
        scan(bad_qual, ["qual"])
    bad_info = "c\t5\t.\tA\tT\t1\t.\tDP=1;MQ=--\tGT:GQ\t0/1:5\n"
    assert scan(bad_info, ["chrom", "qual", "GT"])[0].num_rows == 1                  # no INFO column asked for
    assert scan(bad_info, None, info_fields=[])[0].num_rows == 1                      # no INFO field selected: no builders
    for names in (["DP"], ["MQ"], None):
        with pytest.raises((V.VcfError, ValueError)):
            scan(bad_info, names)                                                     # every entry is typed
    with pytest.raises((V.VcfError, ValueError)):
        scan(bad_info, ["DP"], info_fields=["DP"])                                    # ... selected or not
    # a list is typed but not walked: its elements are parsed only under a key the table has a builder for (:580-611)
    hdr_l = hdr.replace('##INFO=<ID=MQ,Number=1,Type=Float', '##INFO=<ID=MQ,Number=.,Type=Float')
    p2 = tmp_path / "l.vcf"
    p2.write_text(hdr_l + "c\t5\t.\tA\tT\t1\t.\tDP=1;MQ=1.5,--\tGT:GQ\t0/1:5\n")
    o = V.VcfOracle(str(p2), info_fields=["DP"])
    assert o.execute(o.scan(projection=None, filters=[], limit=None, target_partitions=1), 0, 100)[1][0].num_rows == 1
    o = V.VcfOracle(str(p2))                                                          # MQ has a builder, projected or not
    with pytest.raises((V.VcfError, ValueError)):
        o.execute(o.scan(projection=[o.schema.get_field_index("DP")], filters=[], limit=None, target_partitions=1), 0, 100)
    # END sits in front of the entry that does not parse: `end` only walks up to END
    assert scan("c\t5\t.\tAC\tT\t1\t.\tEND=9;MQ=--\tGT:GQ\t0/1:5\n", ["end"])[0].column(0).to_pylist() == [9]
    with pytest.raises((V.VcfError, ValueError)):
        scan("c\t5\t.\tAC\tT\t1\t.\tMQ=--;END=9\tGT:GQ\t0/1:5\n", ["end"])
    bad_gt = "c\t5\t.\tA\tT\t1\t.\tDP=1\tGT:GQ\t0/x:5\n"
    assert scan(bad_gt, None, format_fields=["GQ"])[0].num_rows == 1                  # GT not selected: never walked
    with pytest.raises(V.VcfError):
        scan(bad_gt, None, format_fields=["GT"])
    bad_gq = "c\t5\t.\tA\tT\t1\t.\tDP=1\tGT:GQ\t0/1:5x\n"
    with pytest.raises((V.VcfError, ValueError)):
        scan(bad_gq, None, format_fields=["GT"])                                      # a value of the sample is typed anyway
    assert scan(bad_gq, ["chrom", "DP"])[0].num_rows == 1                             # no FORMAT column asked for
    dup = "c\t5\t.\tA\tT\t1\t.\tDP=1;DP=2;MQ=3\tGT:GQ\t0/1:5\n"
    assert scan(dup, ["MQ"])[0].column(0).to_pylist() == [3.0]
    with pytest.raises(V.VcfError):
        scan(dup, ["DP"])
    for pos in ("-5", "x", "", "12a", "18446744073709551616"):
        with pytest.raises(V.VcfError):
            scan(f"c\t{pos}\t.\tA\tT\t1\t.\tDP=1\tGT:GQ\t0/1:5\n", ["chrom"])
    # a usize that does not fit the u32 columns wraps (`get() as u32`, physical_exec.rs:762, 663-665)
    wrapped = scan("c\t4294967301\t.\tAC\tT\t1\t.\tDP=1\tGT:GQ\t0/1:5\n", ["start", "end"])[0]
    assert wrapped.column(0).to_pylist() == [4] and wrapped.column(1).to_pylist() == [6]
    assert scan("c\t+5\t.\tA\tT\t1\t.\tDP=1\tGT:GQ\t0/1:5\n", ["start"])[0].column(0).to_pylist() == [4]
