"""GPU parity for the VCF path: HIP scan (through the C ABI) vs oracle/vcf_oracle.py -- schema (with metadata),
partition plans, per-partition batches -- on the reference's fixtures, the inline inputs of the reference's own
tests (with the values those tests assert), synthetic config-3 / config-4 style files, and the list UDFs."""
import json
import os
import random
import struct
import subprocess
import sys
import zlib

import numpy as np
import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vcf_cases as C  # noqa: E402


@pytest.fixture(scope="module")
def vo():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vcf_oracle
    return vcf_oracle


def bgzf_compress(data: bytes, block: int = 65280) -> bytes:
    out = []
    for a in range(0, len(data), block):
        raw = data[a:a + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        c = co.compress(raw) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(c) + 25) + c
                   + struct.pack("<II", zlib.crc32(raw) & 0xFFFFFFFF, len(raw)))
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    return b"".join(out)


class GpuTable:
    def __init__(self, pkg, path, info_fields=None, format_fields=None, samples=None, zero_based=True, index_path=None):
        self.p = pkg.VcfTableProvider(path, info_fields, format_fields, None, zero_based, samples, index_path)
        self.schema = self.p.schema()

    def column_names(self):
        return self.schema.names

    def read(self, names=None, filters=(), target_partitions=1, limit=None, batch_size=8192):
        proj = None if names is None else [self.schema.get_field_index(n) for n in names]
        plan = self.p.scan(projection=proj, filters=list(filters), limit=limit, target_partitions=target_partitions)
        cols = {n: [] for n in (names if names is not None else self.schema.names)}
        self.rows = 0
        for p in range(plan.num_partitions()):
            for b in plan.execute(p, batch_size):
                self.rows += b.num_rows
                for n in cols:
                    cols[n].extend(b.column(b.schema.get_field_index(n)).to_pylist())
        return cols


def _schema_equal(a: pa.Schema, b: pa.Schema):
    assert a.names == b.names
    for fa, fb in zip(a, b):
        assert fa.equals(fb, check_metadata=True), (fa, fa.metadata, fb, fb.metadata)
    ma = {k: v for k, v in (a.metadata or {}).items()}
    mb = {k: v for k, v in (b.metadata or {}).items()}
    assert ma.keys() == mb.keys(), (ma.keys(), mb.keys())
    for k in ma:
        va, vb = ma[k], mb[k]
        try:
            assert json.loads(va) == json.loads(vb), k
        except ValueError:
            assert va == vb, k


def _cmp_partition(got, want, ctx, exact_batches=True):
    if exact_batches:
        assert [b.num_rows for b in got] == [b.num_rows for b in want], (ctx, [b.num_rows for b in got][:5], [b.num_rows for b in want][:5])
    tg = pa.Table.from_batches(got) if got else None
    tw = pa.Table.from_batches(want) if want else None
    if tw is None or tg is None:
        assert (tg.num_rows if tg is not None else 0) == (tw.num_rows if tw is not None else 0), ctx
        return
    assert tg.num_rows == tw.num_rows, (ctx, tg.num_rows, tw.num_rows)
    for n in tw.schema.names:
        a, b = tg.column(n).combine_chunks(), tw.column(n).combine_chunks()
        if not a.equals(b):
            la, lb = a.to_pylist(), b.to_pylist()
            for i, (x, y) in enumerate(zip(la, lb)):
                assert x == y or (x != x and y != y), (ctx, n, i, x, y)
            assert a.type == b.type, (ctx, n, a.type, b.type)


def _parity(pkg, vo, path, kw, names=None, filters=(), target=1, limit=None, bs=8192, exact_batches=True, chunk_members=0):
    okw = dict(kw)
    o = vo.VcfOracle(path, **okw)
    g = pkg.VcfTableProvider(path, kw.get("info_fields"), kw.get("format_fields"), None, kw.get("zero_based", True), kw.get("samples"))
    if chunk_members:
        g.set_chunk_members(chunk_members)   # BGZF members per pipeline chunk of the streams below
    _schema_equal(g.schema(), o.schema)
    proj = None if names is None else [o.schema.get_field_index(n) for n in names]
    oplan = o.scan(projection=proj, filters=list(filters), limit=limit, target_partitions=target)
    gplan = g.scan(projection=proj, filters=list(filters), limit=limit, target_partitions=target)
    assert gplan.num_partitions() == o.num_partitions(oplan), (gplan.num_partitions(), o.num_partitions(oplan))
    total = 0
    for p in range(gplan.num_partitions()):
        if oplan["kind"] == "indexed":
            a = oplan["assignments"][p]
            desc = f"{a.total_estimated_bytes}|" + ";".join(
                f"{r.chrom}:{r.start if r.start is not None else ''}-{r.end if r.end is not None else ''}" for r in a.regions)
            assert gplan.partition_desc(p) == desc, (gplan.partition_desc(p), desc)
        got = list(gplan.execute(p, bs))
        osch, want = o.execute(oplan, p, bs)
        if got:
            _schema_equal(got[0].schema, osch)
        _cmp_partition(got, want, (path, names, filters, target, p), exact_batches)
        total += sum(b.num_rows for b in got)
    return total


def test_reference_kats(pkg, tmp_path):
    for mode in ("plain", "bgzf"):
        k = [0]

        def make(text, **kw):
            k[0] += 1
            if mode == "plain":
                p = tmp_path / f"case{k[0]}.vcf"
                p.write_text(text)
            else:
                p = tmp_path / f"case{k[0]}.vcf.gz"
                p.write_bytes(bgzf_compress(text.encode(), 97))   # tiny blocks: lines span BGZF members
            return GpuTable(pkg, str(p), **kw)
        C.check_reference_kats(make)


@pytest.mark.parametrize("name,per", [("multi_chrom.vcf.gz", 500), ("multi_chrom_large.vcf.gz", 5000)])
def test_fixture_indexed_parity(pkg, vo, name, per):
    path = os.path.join(G, name)
    for tp in (1, 2, 3, 4, 8):
        assert _parity(pkg, vo, path, {}, target=tp) == 2 * per                                  # indexed_read_test.rs:136-152
        assert _parity(pkg, vo, path, {}, names=["chrom"], filters=[("chrom", "=", "21")], target=tp) == per   # :99-112
        assert _parity(pkg, vo, path, {}, names=["chrom", "start"], filters=[("chrom", "in", ["21", "22"])], target=tp) == 2 * per
    n = _parity(pkg, vo, path, {}, names=["chrom"], target=4,
                filters=[("chrom", "=", "21"), ("start", ">=", 5009999), ("start", "<=", 5029999)])           # :197-216
    assert 0 < n < per
    _parity(pkg, vo, path, {}, names=[], target=3, filters=[("chrom", "=", "22")])
    _parity(pkg, vo, path, {}, names=["qual", "filter", "AF", "DB", "DP", "end", "id", "alt", "ref"], target=2, bs=100)
    _parity(pkg, vo, path, {}, names=["chrom", "qual"], filters=[("qual", ">=", 50.0)], target=2)            # :156-170 (qual passes through)
    _parity(pkg, vo, path, {}, filters=[("chrom", "=", "21"), ("end", "<=", 5010000), ("id", "!=", "rs3")], target=2)
    _parity(pkg, vo, path, {}, filters=[("chrom", "=", "nope")], target=2)
    assert _parity(pkg, vo, path, {}, names=["chrom"], filters=[("chrom", "=", "21")], limit=5) == 5          # limit tests :233-262
    assert _parity(pkg, vo, path, {}, names=["chrom", "start"], filters=[("chrom", "=", "22")], limit=1) == 1
    assert _parity(pkg, vo, path, {}, names=["chrom"], filters=[("chrom", "=", "21")], limit=9999) == per
    assert _parity(pkg, vo, path, {"zero_based": False, "info_fields": []}, names=[], target=4,
                   filters=[("chrom", "=", "21"), ("start", "=", 5000100)]) == 1                              # :222-232
    g = pkg.VcfTableProvider(path)
    assert g.scan(filters=[("chrom", "=", "21"), ("start", "=", 5000100), ("start", ">", 5000100)]).num_partitions() == 0
    assert g.scan(limit=0).num_partitions() == 0
    assert g.supports_filters_pushdown([("chrom", "=", "21"), ("qual", ">=", 5.0), ("DB", "=", 1), ("AF", "=", 1.0)]) == \
        ["Inexact", "Inexact", "Unsupported", "Unsupported"]
    assert g.scan(projection=[0, 1]).display() == "VcfExec: projection=[chrom, start]"


def test_fixture_sequential_parity(pkg, vo, tmp_path):
    # no index -> one sequential partition (table_provider.rs:1410-1462)
    src = os.path.join(G, "multi_chrom.vcf.gz")
    dst = tmp_path / "noindex.vcf.gz"
    dst.write_bytes(open(src, "rb").read())
    assert _parity(pkg, vo, str(dst), {}, target=4) == 1000
    assert _parity(pkg, vo, str(dst), {}, names=["chrom", "start"], limit=3, target=4) == 3
    plain = tmp_path / "plain.vcf"
    plain.write_bytes(vo.bgzf_decompress(open(src, "rb").read()))
    assert _parity(pkg, vo, str(plain), {}, target=2, bs=77) == 1000


def test_real_multisample_fixture(pkg, vo):
    # format_columns_test.rs:378-398: 2504 samples, AD (Number=.) and PL (Number=G) -> List<List<Int32>>
    path = os.path.join(G, "head_106667_tail_6.vcf")
    kw = {"info_fields": [], "format_fields": ["GT", "AD", "DP", "GQ", "PL"]}
    assert _parity(pkg, vo, path, kw, names=["chrom", "start", "genotypes"], exact_batches=False) == 6
    kw = {"format_fields": ["GT", "DP"], "samples": ["HG00100", "HG00096", "nobody"]}
    assert _parity(pkg, vo, path, kw, exact_batches=False) == 6


def _synth(tmp_path, *args):
    exe = os.path.join(ROOT, "tools", "_build", "synth_vcf")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")], stdout=subprocess.DEVNULL)
    out = subprocess.check_output([exe] + [str(a) for a in args])
    return json.loads(out)


def test_synth_sites_parity(pkg, vo, tmp_path):
    path = str(tmp_path / "sites.vcf.gz")
    meta = _synth(tmp_path, "sites", path, 20000, 11)
    for tp in (1, 4, 16):
        assert _parity(pkg, vo, path, {}, target=tp, bs=1000) == 20000
    n = _parity(pkg, vo, path, {}, filters=[("chrom", "=", "chr1")], target=8)                   # config 3 predicate
    assert n == meta["n_lines_chr1"]
    _parity(pkg, vo, path, {"zero_based": False}, names=["chrom", "start", "end", "AF", "RSRC", "DB", "VQSLOD"],
            filters=[("chrom", "in", ["chr2", "chr21"]), ("start", ">", 1000000)], target=5)
    _parity(pkg, vo, path, {"info_fields": ["AF", "VT"]}, filters=[("chrom", "=", "chr7"), ("start", "between", (20000000, 90000000))], target=3)


def test_synth_samples_parity(pkg, vo, tmp_path):
    path = str(tmp_path / "samples.vcf.gz")
    _synth(tmp_path, "samples", path, 400, 70, 5)
    assert _parity(pkg, vo, path, {}, target=1, exact_batches=True) == 400
    assert _parity(pkg, vo, path, {}, target=6) == 400
    _parity(pkg, vo, path, {"format_fields": ["GQ", "GT"], "samples": ["S00070", "S00001", "S00033"]}, names=["start", "genotypes"],
            filters=[("chrom", "=", "chr2")], target=2)
    _parity(pkg, vo, path, {}, names=["chrom", "AF"], target=2)   # FORMAT not projected: plain batch size


def test_thousand_sample_parity(pkg, vo, tmp_path):
    """BASELINE.json config 4's shape -- 1000 samples, FORMAT GT:GQ:DP -- value by value against the oracle (the
    reference's adaptive batch sizing makes batch boundaries irreproducible there, so rows are compared per partition),
    plus the list UDFs on the GQ / DP lists of those rows."""
    path = str(tmp_path / "s1000.vcf.gz")
    _synth(tmp_path, "samples", path, 90, 1000, 23)
    assert _parity(pkg, vo, path, {}, target=1, exact_batches=False) == 90
    assert _parity(pkg, vo, path, {}, target=4, exact_batches=False) == 90
    _parity(pkg, vo, path, {"format_fields": ["DP", "GT"]}, names=["chrom", "start", "genotypes"], target=3, exact_batches=False)
    g = pkg.VcfTableProvider(path)
    o = vo.VcfOracle(path)
    gi = g.schema().get_field_index("genotypes")
    rows = pa.Table.from_batches(list(g.scan(projection=[gi]).execute(0, 8192)))
    st = rows.column("genotypes").combine_chunks()
    gq, dp = st.field("GQ"), st.field("DP")
    assert len(gq) == 90 and all(len(x) == 1000 for x in gq.to_pylist())
    want_avg = [None if not [v for v in r if v is not None] else sum(float(v) for v in r if v is not None) / len([v for v in r if v is not None])
                for r in gq.to_pylist()]
    assert pkg.list_avg(gq).to_pylist() == want_avg                        # udfs.rs:67-110, sequential f64 accumulation
    assert pkg.list_gte(dp, 10).to_pylist() == [[None if v is None else v >= 10 for v in r] for r in dp.to_pylist()]
    assert pkg.list_lte(dp, 200).to_pylist() == [[None if v is None else v <= 200 for v in r] for r in dp.to_pylist()]


def test_list_udfs_host(pkg, vo):
    rnd = random.Random(3)
    L = pa.list_(pa.field("item", pa.int32(), True))
    rows = []
    for _ in range(3000):
        r = rnd.random()
        if r < 0.05:
            rows.append(None)
        else:
            rows.append([None if rnd.random() < 0.1 else rnd.randint(-50, 300) for _ in range(rnd.randint(0, 40))])
    a = pa.array(rows, type=L)
    assert pkg.list_avg(a).equals(vo.list_avg(a))
    assert pkg.list_gte(a, 20).equals(vo.list_gte(a, 20))
    assert pkg.list_lte(a, 100).equals(vo.list_lte(a, 100))
    assert pkg.list_avg(a.slice(17, 1000)).equals(vo.list_avg(a.slice(17, 1000)))
    F = pa.list_(pa.field("item", pa.float32(), True))
    fr = [None if rnd.random() < 0.05 else [None if rnd.random() < 0.1 else rnd.uniform(-1e3, 1e3) * 10 ** rnd.randint(-6, 6)
                                             for _ in range(rnd.randint(0, 30))] for _ in range(2000)]
    f = pa.array(fr, type=F)
    assert pkg.list_avg(f).equals(vo.list_avg(f))
    assert pkg.list_gte(f, 0.5).equals(vo.list_gte(f, 0.5))
    # udfs.rs:1063-1110 vectors
    gq = pa.array([[30, 20, 10], [5, None, 15]], type=L)
    assert pkg.list_avg(gq).to_pylist() == [20.0, 10.0]
    assert pkg.list_gte(gq, 15).to_pylist() == [[True, True, False], [False, None, True]]


def test_list_and_and_set_gts(pkg, vo):
    rnd = random.Random(8)
    B = pa.list_(pa.field("item", pa.bool_(), True))
    G = pa.list_(pa.field("item", pa.utf8(), True))

    def blist(n, maxlen):
        return pa.array([None if rnd.random() < 0.05 else [None if rnd.random() < 0.15 else rnd.random() < 0.5
                                                          for _ in range(rnd.randint(0, maxlen))] for _ in range(n)], type=B)
    a, b = blist(1500, 90), blist(1500, 90)
    assert pkg.list_and(a, b).equals(vo.list_and(a, b))
    assert pkg.list_and(a.slice(3, 700), b.slice(3, 700)).equals(vo.list_and(a.slice(3, 700), b.slice(3, 700)))
    gts = ["0/0", "0/1", "1|1", "./.", ".", "10/11", ""]
    gt = pa.array([None if rnd.random() < 0.05 else [None if rnd.random() < 0.1 else rnd.choice(gts) for _ in range(rnd.randint(0, 90))]
                   for _ in range(1500)], type=G)
    assert pkg.vcf_set_gts(gt, a, "./.").equals(vo.vcf_set_gts(gt, a, "./."))
    assert pkg.vcf_set_gts(gt, b, ".").equals(vo.vcf_set_gts(gt, b, "."))
    # udfs.rs:1112-1162 vectors
    L = pa.list_(pa.field("item", pa.int32(), True))
    gq = pa.array([[30, 20, 10], [5, None, 15]], type=L)
    dp = pa.array([[50, 30, 20], [10, 200, 100]], type=L)
    g2 = pa.array([["0/1", "1/1", "0/0"], ["./.", "0/1", "1/1"]], type=G)
    assert pkg.vcf_set_gts(g2, pkg.list_gte(gq, 15), "./.").to_pylist() == [["0/1", "1/1", "./."], ["./.", "0/1", "1/1"]]
    assert pkg.list_and(pkg.list_gte(gq, 10), pkg.list_lte(dp, 100)).to_pylist() == [[True, True, True], [False, False, True]]


def test_list_udfs_device_resident(pkg, vo, tmp_path):
    path = str(tmp_path / "samples.vcf.gz")
    _synth(tmp_path, "samples", path, 300, 120, 9)
    o = vo.VcfOracle(path)
    g = pkg.VcfTableProvider(path)
    gi = o.schema.get_field_index("genotypes")
    oplan = o.scan(projection=[gi])
    gplan = g.scan(projection=[gi])
    _, want = o.execute(oplan, 0)
    t = pa.Table.from_batches(want)
    geno = t.column("genotypes").combine_chunks()
    gq, dp = geno.field("GQ"), geno.field("DP")
    avg = vo.list_avg(gq)
    r = gplan.execute_device_udf(0, "GQ", "list_avg")["udf"]
    assert r["n_rows"] == 300 and r["count_a"] == len(avg) - avg.null_count
    assert r["sum"] == sum(v for v in avg.to_pylist() if v is not None)
    ge = vo.list_gte(dp, 100)
    r = gplan.execute_device_udf(0, "DP", "list_gte", 100)["udf"]
    flat = [x for row in ge.to_pylist() for x in row]
    assert r["count_a"] == sum(1 for x in flat if x) and r["count_b"] == sum(1 for x in flat if x is None)
    r = gplan.execute_device_udf(0, "DP", "list_lte", 30)["udf"]
    flat = [x for row in vo.list_lte(dp, 30).to_pylist() for x in row]
    assert r["count_a"] == sum(1 for x in flat if x)


def test_float_parse_fuzz(pkg, tmp_path):
    """decimal -> f32 on the device must equal the correctly rounded value (Rust `str::parse::<f32>`)."""
    rnd = random.Random(12)
    vals = ["0", "0.0", "1", "-1", "0.1", "0.3", "1e-5", "1E5", "3.4028235e38", "1.17549435e-38", "16777217", "0.998595",
            "1e39", "1e-60", "123456789012345678901234567890",
            "1e-40", "3.1e-42", "1.4e-45", "1e-46", "7e-46", "7.1e-46", "1.1754942e-38", "-2.5e-44", "0.00000000000000000000000000000000000000000314", "0.000000000000000000001", "+5.5", "5.", ".5", "nan", "inf", "-Infinity"]
    for _ in range(6000):
        kind = rnd.random()
        if kind < 0.4:
            vals.append(f"{rnd.uniform(0, 1):.{rnd.randint(1, 9)}g}")
        elif kind < 0.7:
            vals.append(f"{rnd.uniform(-1e4, 1e4) * 10 ** rnd.randint(-12, 12):.{rnd.randint(1, 17)}e}")
        elif kind < 0.85:
            vals.append(repr(float(np.float32(rnd.uniform(-10, 10)))))
        else:
            vals.append(repr(rnd.uniform(-1, 1) * 10 ** rnd.randint(-30, 30)))
    lines = ["##fileformat=VCFv4.3", "##INFO=<ID=X,Number=1,Type=Float,Description=\"x\">",
             "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"]
    for i, v in enumerate(vals):
        lines.append(f"c\t{i + 1}\t.\tA\tT\t{v}\t.\tX={v}")
    p = tmp_path / "floats.vcf"
    p.write_text("\n".join(lines) + "\n")
    t = GpuTable(pkg, str(p))
    r = t.read(["qual", "X"])
    want = [float(np.float32(v)) for v in vals]
    for v, q, x, w in zip(vals, r["qual"], r["X"], want):
        if w != w:
            assert q != q and x != x, v
        else:
            assert q == w and x == w, (v, q, x, w)


def test_float_literals_on_rounding_boundaries(pkg, vo, tmp_path):
    """Literals that sit exactly on the midpoint of two f32 values (26 to 118 significant digits), a hair above and below them,
    the overflow boundary and 40-digit literals: the device's estimate cannot decide these and its exact integer comparison
    (k_f32_fix) must -- ties to even, like Rust's parser.  Expected values come from exact rational arithmetic in the oracle."""
    from decimal import Decimal, getcontext
    from fractions import Fraction
    import struct
    getcontext().prec = 400

    def f32(bits):
        return struct.unpack("<f", struct.pack("<I", bits))[0]

    rnd = random.Random(5)
    vals = []
    bit_patterns = [0x3F800000, 0x3F800001, 0x00000001, 0x00000002, 0x007FFFFF, 0x00800000, 0x4B000000, 0x7F7FFFFE, 0x3DCCCCCC, 0x0A4FB11E]
    bit_patterns += [rnd.randrange(1, 0x7F7FFFFF) for _ in range(120)]
    for bits in bit_patterns:
        lo, hi = f32(bits), f32(bits + 1)
        mid = (Fraction(lo) + Fraction(hi)) / 2
        d = Decimal(mid.numerator) / Decimal(mid.denominator)          # exact: the midpoint is a dyadic rational
        exact = format(d, "f") if 1e-6 < mid < 1e21 else "%se%d" % (str(d.scaleb(-d.adjusted())), d.adjusted())
        assert Fraction(Decimal(exact)) == mid
        frac = "." not in exact.split("e")[0] and "e" not in exact
        for lit in (exact, "-" + exact):
            vals.append(lit)
        m, _, ex = exact.partition("e")
        if "." not in m:
            m += "."
        for tail in ("0000000000000000000000001", "1"):
            vals.append(m + tail + ("e" + ex if ex else ""))             # a hair above the tie: rounds up
        # a hair below: the last digit lowered by one and nines appended
        body = m.rstrip(".")
        if body[-1] != "0":
            vals.append(body[:-1] + str(int(body[-1]) - 1) + "9999999999999999999999999999" + ("e" + ex if ex else ""))
    vals += ["340282356779733661637539395458142568448", "340282356779733661637539395458142568447.9999", "340282356779733661637539395458142568448.0001",
             "3.4028235677973366e38", "3.4028235677973365e38", "0.1000000000000000055511151231257827021181583404541015625",
             "16777217.000000000000000000000000000000000001", "16777217", "16777216.999999999999999999999999999999",
             "1.00000005960464477539062500000000000000000000000000001", "1.000000059604644775390625", "0.999999970197677612304687500"]
    lines = ["##fileformat=VCFv4.3", "##INFO=<ID=X,Number=1,Type=Float,Description=\"x\">", "##INFO=<ID=L,Number=.,Type=Float,Description=\"l\">",
             "##FORMAT=<ID=GL,Number=1,Type=Float,Description=\"g\">",
             "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2"]
    for i, v in enumerate(vals):
        lines.append(f"c\t{i + 1}\t.\tA\tT\t{v}\t.\tX={v};L=1.5,{v}\tGL\t{v}\t0.25")
    p = tmp_path / "ties.vcf"
    p.write_text("\n".join(lines) + "\n")
    t = GpuTable(pkg, str(p), info_fields=["X", "L"], format_fields=["GL"])
    r = t.read(["qual", "X", "L", "genotypes"])
    want = [vo.parse_f32(v) for v in vals]
    n_inf = 0
    for i, (v, w) in enumerate(zip(vals, want)):
        gl = r["genotypes"][i]["GL"]
        got = (r["qual"][i], r["X"][i], r["L"][i][1], gl[0])
        assert all(g == w for g in got), (v, got, w)
        assert r["L"][i][0] == 1.5 and gl[1] == 0.25
        n_inf += w == float("inf")
    assert n_inf >= 3


def test_vcf_errors_are_loud(pkg, tmp_path):
    """Malformed input and unsupported encodings raise (the reference yields DataFusionError::Execution);
    nothing is silently skipped or guessed."""
    import gzip
    hdr = ("##fileformat=VCFv4.3\n##INFO=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n"
           "##INFO=<ID=S,Number=1,Type=String,Description=\"s\">\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")

    def run(body, name="x.vcf", **kw):
        p = tmp_path / name
        p.write_text(hdr + body)
        t = GpuTable(pkg, str(p), **kw)
        return t.read()
    assert run("c\t5\t.\tA\tT\t.\t.\tDP=3\n")["DP"] == [3]
    for body, msg in (("c\t5\t.\tA\tT\t.\t.\tDP=x\n", "invalid integer"),
                      ("c\t5\t.\tA\tT\t.\t.\tDP=1;DP=2\n", "duplicate INFO key"),
                      ("c\t5\t.\tA\tT\t.\t.\tS=a%C3%28\n", "invalid UTF-8"),      # an escape that does not decode to UTF-8
                      ("c\t5\t.\tA\tT\t.\t.\tS=a%ED%A0%80\n", "invalid UTF-8"),   # a surrogate
                      ("c\t5\t.\tA\tT\tbad\t.\tDP=1\n", "qual"),
                      ("c\t0\t.\tA\tT\t.\t.\tDP=1\n", "Missing variant start"),
                      ("c\tx\t.\tA\tT\t.\t.\tDP=1\n", "position"),
                      ("c\t5\t.\tA\n", "fewer than 8"),
                      ("c\t5\t.\tA\tT\t.\t.\tDP=1\n\nc\t6\t.\tA\tT\t.\t.\tDP=1\n", "blank line")):
        with pytest.raises(pkg.BioscanError) as ei:
            run(body)
        assert msg.lower() in str(ei.value).lower(), (body, str(ei.value))
    # plain gzip (no BGZF block structure) is refused at open; a CSI index only at execute (as the reference's tabix reader does)
    gz = tmp_path / "plain.vcf.gz"
    gz.write_bytes(gzip.compress((hdr + "c\t5\t.\tA\tT\t.\t.\tDP=3\n").encode()))
    with pytest.raises(pkg.BioscanError) as ei:
        pkg.VcfTableProvider(str(gz))
    assert "not BGZF" in str(ei.value)
    bg = tmp_path / "b.vcf.gz"
    bg.write_bytes(bgzf_compress((hdr + "c\t5\t.\tA\tT\t.\t.\tDP=3\n").encode()))
    (tmp_path / "b.vcf.gz.csi").write_bytes(b"CSI\x01")
    prov = pkg.VcfTableProvider(str(bg))  # unreadable index + no ##contig lines: no regions, so the sequential fallback runs
    assert sum(b.num_rows for b in prov.scan().execute(0, 8192)) == 1
    # an INFO tag that the header does not define is refused (the reference unwraps the lookup and panics)
    with pytest.raises(pkg.BioscanError):
        pkg.VcfTableProvider(str(tmp_path / "x.vcf"), ["NOPE"])


def test_percent_decoding(pkg, vo, tmp_path):
    """noodles percent-decodes INFO / FORMAT string values (oracle header); the device copy does the same."""
    text = ("##fileformat=VCFv4.3\n##INFO=<ID=S,Number=1,Type=String,Description=\"s\">\n"
            "##INFO=<ID=L,Number=.,Type=String,Description=\"l\">\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\n"
            "##FORMAT=<ID=FT,Number=1,Type=String,Description=\"f\">\n"
            "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\n"
            "c\t5\t.\tA\tT\t.\t.\tS=a%3Bb%3Dc;L=x%2Cy,.,%25z\tGT:FT\t0/1:p%3Aq\t1/1:.\n"
            "c\t6\t.\tA\tT\t.\t.\tS=plain;L=%zz,100%\tGT:FT\t0/0:ok\t./.:%41\n"
            "c\t7\t.\tA\tT\t.\t.\tS=caf%C3%A9;L=%E2%82%AC,%F0%9F%A7%AC\tGT:FT\t0/0:%C3%9F\t./.:x\n")
    p = tmp_path / "pct.vcf"
    p.write_text(text)
    assert _parity(pkg, vo, str(p), {}, exact_batches=False) == 3
    t = GpuTable(pkg, str(p))
    r = t.read(["S", "L", "genotypes"])
    # escapes >= 0x80 are fine as long as the decoded value is UTF-8 (noodles: percent_decode(..).decode_utf8())
    assert r["S"] == ["a;b=c", "plain", "caf\u00e9"] and r["L"] == [["x,y", None, "%z"], ["%zz", "100%"], ["\u20ac", "\U0001f9ec"]]
    assert r["genotypes"][2]["FT"] == ["\u00df", "x"]
    assert r["genotypes"][0]["FT"] == ["p:q", None] and r["genotypes"][1]["FT"] == ["ok", "A"]


def test_vcf_multi_gpu_sharding_reproduces_single_gpu_order(pkg, vo, tmp_path):
    """SURVEY 8e for VCF: TBI partitions sharded in order across 2/4/8 ranks (simulated on one GPU); concatenating
    the ranks' rows in rank order reproduces the single-GPU row order."""
    path = str(tmp_path / "sites.vcf.gz")
    _synth(tmp_path, "sites", path, 12000, 21)
    g = pkg.VcfTableProvider(path)
    ci, si = g.schema().get_field_index("chrom"), g.schema().get_field_index("start")

    def rows_of(plan, parts):
        out = []
        for p in parts:
            for b in plan.execute(p):
                out += list(zip(b.column("chrom").to_pylist(), b.column("start").to_pylist()))
        return out
    for world in (2, 4, 8):
        plan = g.scan(projection=[ci, si], target_partitions=world * 3)
        n = plan.num_partitions()
        single = rows_of(plan, range(n))
        shards = pkg.shard_partitions_in_order([plan.partition_estimated_bytes(p) for p in range(n)], world)
        assert sum(len(s) for s in shards) == n
        multi = []
        for rank in range(world):
            multi += rows_of(plan, shards[rank])
        assert multi == single and len(single) == 12000


def test_csi_and_unreadable_indexes(pkg, vo, tmp_path):
    """A CSI companion only contributes `bio.vcf.contigs.indexed` (indexed_read_test.rs:380-417); an index the tabix
    reader rejects is soft at open and loud at execute (table_provider.rs:1012-1024, physical_exec.rs:2766-2768)."""
    path = os.path.join(G, "multi_chrom_csi.vcf.gz")
    o = vo.VcfOracle(path)
    g = pkg.VcfTableProvider(path)
    _schema_equal(g.schema(), o.schema)
    assert sorted(json.loads(g.schema().metadata[b"bio.vcf.contigs.indexed"])) == ["21", "22"]
    for target, filters in ((4, []), (1, [("chrom", "=", "21")])):
        gp, op = g.scan(filters=filters, target_partitions=target), o.scan(filters=filters, target_partitions=target)
        assert gp.num_partitions() == o.num_partitions(op)
        with pytest.raises(vo.VcfError):
            o.execute(op, 0)
        with pytest.raises(RuntimeError, match="Failed to open indexed VCF"):
            list(gp.execute(0, 8192))
    # no index: the sequential scan of the same file
    kw = dict(index_path=None)
    o2 = vo.VcfOracle(path, **kw)
    g2 = pkg.VcfTableProvider(path, None, None, None, True, None, "")
    _schema_equal(g2.schema(), o2.schema)
    got = list(g2.scan().execute(0, 8192))
    _, want = o2.execute(o2.scan())
    _cmp_partition(got, want, "csi-sequential")
    assert sum(b.num_rows for b in got) == 1000
    # garbage given as the index
    bad = tmp_path / "bad.tbi"
    bad.write_bytes(bgzf_compress(b"not an index at all"))
    src = os.path.join(G, "multi_chrom.vcf.gz")
    o3 = vo.VcfOracle(src, index_path=str(bad))
    g3 = pkg.VcfTableProvider(src, None, None, None, True, None, str(bad))
    _schema_equal(g3.schema(), o3.schema)
    assert b"bio.vcf.contigs.indexed" not in (g3.schema().metadata or {})
    gp, op = g3.scan(target_partitions=2), o3.scan(target_partitions=2)
    assert gp.num_partitions() == o3.num_partitions(op)
    with pytest.raises(vo.VcfError):
        o3.execute(op, 0)
    with pytest.raises(RuntimeError, match="Failed to open indexed VCF"):
        list(gp.execute(0, 8192))
    missing = str(tmp_path / "nope.tbi")
    g4 = pkg.VcfTableProvider(src, None, None, None, True, None, missing)
    o4 = vo.VcfOracle(src, index_path=missing)
    _schema_equal(g4.schema(), o4.schema)
    with pytest.raises(RuntimeError, match="Failed to open indexed VCF"):
        list(g4.scan().execute(0, 8192))


def _ms_vcf(cells_rows, fmt="GT:GQ:DP", extra_formats=()):
    ns = len(cells_rows[0])
    hdr = ["##fileformat=VCFv4.3", "##contig=<ID=c1,length=1000000>",
           '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">', '##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="q">',
           '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">', '##FORMAT=<ID=PL,Number=G,Type=Integer,Description="p">',
           '##FORMAT=<ID=FT,Number=1,Type=String,Description="f">', '##FORMAT=<ID=AF,Number=1,Type=Float,Description="a">']
    hdr += list(extra_formats)
    hdr.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"S{i}" for i in range(ns)))
    lines = []
    for k, row in enumerate(cells_rows):
        f = fmt[k] if isinstance(fmt, list) else fmt
        lines.append(f"c1\t{100 + k}\t.\tA\tT\t.\t.\t.\t{f}\t" + "\t".join(row))
    return "\n".join(hdr + lines) + "\n"


def test_format_cell_corners(pkg, vo, tmp_path):
    """FORMAT cells the synthetic generator never writes: short and long cells (the 16-byte fast path and the byte-wise
    path), signs, missing values, trailing fields left out, empty trailing sub-fields, phasing marks in front,
    multi-digit alleles, keys in any order and keys that are not selected."""
    rng = random.Random(77)
    gts = ["0/1", "1|0", ".", "./.", "0", "10/2", "|0|1", "/1", "1/2/3", ".|.", "0|0|0|0", "12|345"]
    ints = ["0", "7", "99", "250", ".", "-5", "+7", "2147483647", "-2147483648", "007"]
    pls = ["0,30,300", ".", "0,255,255,0,12,1000", "1"]
    fts = ["PASS", ".", "q10;s50", "a"]
    afs = ["0.5", ".", "1e-3", "12.25"]
    rows, fmts = [], []
    for k in range(400):
        keys = rng.choice([["GT", "GQ", "DP"], ["GT", "DP", "GQ"], ["GQ", "GT"], ["GT", "GQ", "DP", "PL"], ["GT", "FT", "GQ", "PL", "DP"],
                           ["DP"], ["GT", "AF", "DP"], ["PL", "GT", "GQ"]])
        row = []
        for s in range(6):
            vals = {"GT": rng.choice(gts), "GQ": rng.choice(ints), "DP": rng.choice(ints), "PL": rng.choice(pls), "FT": rng.choice(fts),
                    "AF": rng.choice(afs)}
            cell = [vals[x] for x in keys]
            cut = rng.choice([None, None, None, 1, 2])          # trailing fields dropped
            if cut is not None:
                cell = cell[:max(1, min(cut, len(cell)))]
            txt = ":".join(cell)
            if rng.random() < 0.05:
                txt = "."
            row.append(txt)
        rows.append(row)
        fmts.append(":".join(keys))
    p = tmp_path / "cells.vcf"
    p.write_text(_ms_vcf(rows, fmts))
    for ff in (["GT", "GQ", "DP"], ["DP", "PL", "GT"], ["GT", "FT", "AF", "GQ", "DP", "PL"], ["GQ"]):
        assert _parity(pkg, vo, str(p), dict(format_fields=ff)) == 400
    pz = tmp_path / "cells.vcf.gz"
    pz.write_bytes(bgzf_compress(p.read_bytes(), 211))
    assert _parity(pkg, vo, str(pz), dict(format_fields=["GT", "GQ", "DP"], index_path=None)) == 400


@pytest.mark.parametrize("cell,fmt", [
    ("0/x:5:5", "GT:GQ:DP"), ("1/:5:5", "GT:GQ:DP"), ("a:5:5", "GT:GQ:DP"), ("1//2:5:5", "GT:GQ:DP"),
    ("0/1:1x:5", "GT:GQ:DP"), ("0/1:--1:5", "GT:GQ:DP"), ("0/1::5", "GT:GQ:DP"), ("0/1:5:2147483648", "GT:GQ:DP"),
    ("0/1:5:-", "GT:GQ:DP"), ("0/1:5:5:", "GT:GQ:DP:GQ2"),
    ("0/1:99999999999999999999:5", "GT:GQ:DP"), ("0/1:5:5:0,1x,3:abcdefghij", "GT:GQ:DP:PL:FT"),
])
def test_format_cell_errors_are_loud_on_both_sides(pkg, vo, tmp_path, cell, fmt):
    extra = ['##FORMAT=<ID=GQ2,Number=1,Type=Integer,Description="x">']
    p = tmp_path / "bad.vcf"
    p.write_text(_ms_vcf([["0/1:1:1", cell]], fmt, extra))
    ff = fmt.split(":")
    o = vo.VcfOracle(str(p), format_fields=ff)
    with pytest.raises(Exception):
        o.execute(o.scan())
    g = pkg.VcfTableProvider(str(p), None, ff)
    with pytest.raises(pkg.BioscanError):
        list(g.scan().execute(0, 8192))


def test_gt_with_leading_zero_alleles_is_rendered_again(pkg, vo, tmp_path):
    """noodles parses allele indices as integers and the reference renders them again (physical_exec.rs:1675-1694): "01/1"
    comes out as "1/1", "00|007" as "0|7" -- on the device the cell is sized for the rendered form and written by
    k_gt_render (r03 refused such a file)."""
    p = tmp_path / "lz.vcf"
    p.write_text(_ms_vcf([["0/1:1:1", "01/1:5:5", "00|007:1:2"], ["./000:3:3", "|010/0:4:4", "10/020/3:5:6"]]))
    for kw in ({"format_fields": ["GT", "GQ", "DP"]}, {"format_fields": ["GT"], "samples": ["S2", "S1"]}):
        assert _parity(pkg, vo, str(p), kw, exact_batches=False) == 2
    g = pkg.VcfTableProvider(str(p), None, ["GT"])
    t = pa.Table.from_batches(list(g.scan().execute(0, 8192)))
    assert t.column("genotypes").to_pylist()[0]["GT"] == ["0/1", "1/1", "0|7"]
    assert t.column("genotypes").to_pylist()[1]["GT"] == ["./0", "10/0", "10/20/3"]
    # single-sample source: the GT column of the one sample
    q = tmp_path / "lz1.vcf"
    q.write_text(_ms_vcf([["007/01:1:1"], ["1|02:2:2"]]))
    assert _parity(pkg, vo, str(q), {"format_fields": ["GT", "DP"]}) == 2
    # GT as the eleventh selected key (the cell kernel parses its first eight keys itself: GT is always one of them --
    # tools/fuzz_vcf_parity.py seed 37 found "0000" coming out as written when it was not)
    extra = [f'##FORMAT=<ID=K{i},Number=1,Type=Integer,Description="k">' for i in range(10)]
    fmt = ":".join(f"K{i}" for i in range(10)) + ":GT"
    ints = ":".join(str(i) for i in range(10))
    for rows in ([[f"{ints}:0000/01", f"{ints}:1|000"]], [[f"{ints}:|007/0"]]):
        r = tmp_path / "lz11.vcf"
        r.write_text(_ms_vcf(rows, fmt, extra))
        assert _parity(pkg, vo, str(r), {"format_fields": [f"K{i}" for i in range(10)] + ["GT"]}, exact_batches=False) == 1


def test_values_without_a_column_are_typed_all_the_same(pkg, vo, tmp_path):
    """noodles types every INFO entry (`info.iter(header)`, physical_exec.rs:561-571) and every value of a selected sample
    (`sample.iter(header)`, :1661-1666) as it walks them: a scalar that does not parse under a key the scan has NO column for is
    the record's error all the same -- once some INFO / FORMAT column is asked for.  A list or a genotype stays unwalked unless
    the table has a builder for its key (:580-611, :1668-1760); `end` walks INFO up to END only, and not at all for a
    single-base substitution (:646-667).  Each case: both sides refuse, or both read the same rows."""
    hdr = ("##fileformat=VCFv4.3\n##contig=<ID=c,length=100000>\n"
           '##INFO=<ID=DP,Number=1,Type=Integer,Description="d">\n##INFO=<ID=MQ,Number=1,Type=Float,Description="m">\n'
           '##INFO=<ID=LST,Number=.,Type=Integer,Description="l">\n##INFO=<ID=DB,Number=0,Type=Flag,Description="f">\n'
           '##INFO=<ID=CH,Number=1,Type=Character,Description="c">\n##INFO=<ID=NOTE,Number=1,Type=String,Description="n">\n'
           '##INFO=<ID=END,Number=1,Type=Integer,Description="e">\n'
           '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">\n##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="q">\n'
           '##FORMAT=<ID=XF,Number=1,Type=Float,Description="x">\n##FORMAT=<ID=PL,Number=G,Type=Integer,Description="p">\n'
           '##FORMAT=<ID=FC,Number=1,Type=Character,Description="c">\n##FORMAT=<ID=FT,Number=1,Type=String,Description="t">\n'
           "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS0\tS1\n")
    good = "c\t5\t.\tA\tT\t1\t.\tDP=1;MQ=2.5\tGT:GQ\t0/1:5\t1/1:6\n"

    def both(body, kw, names, expect):
        p = tmp_path / "t.vcf"
        p.write_text(hdr + good + body)
        if expect == "read":
            assert _parity(pkg, vo, str(p), kw, names=names, exact_batches=False) == 2, (body, kw, names)
            return
        o = vo.VcfOracle(str(p), **kw)
        proj = None if names is None else [o.schema.get_field_index(n) for n in names]
        with pytest.raises((vo.VcfError, ValueError)):
            o.execute(o.scan(projection=proj, filters=[], limit=None, target_partitions=1), 0, 100)
        g = pkg.VcfTableProvider(str(p), kw.get("info_fields"), kw.get("format_fields"), None, True, kw.get("samples"))
        with pytest.raises(pkg.BioscanError):
            list(g.scan(projection=proj, filters=[], limit=None, target_partitions=1).execute(0, 100))

    rec = lambda info, fmt="GT:GQ", s0="0/1:5", s1="1/1:6", ref="A": f"c\t9\t.\t{ref}\tT\t1\t.\t{info}\t{fmt}\t{s0}\t{s1}\n"
    # ---- INFO: a scalar of a key without a column
    for info in ("DP=1;MQ=--", "MQ=--;DP=1", "DP=1x", "DB=1", "DP=1;CH=xy", "NOTE=%FF;DP=1", "UNDECLARED=%FF", "DP=1;;MQ=1e"):
        both(rec(info), {"info_fields": ["DP"]}, ["chrom", "DP"], "refuse")          # MQ / DB / CH / NOTE are not table fields
        both(rec(info), {}, ["chrom", "LST"], "refuse")                               # ... or are, but not projected
        both(rec(info), {}, ["chrom", "start", "genotypes"], "read")                  # no INFO column: INFO is never walked
    both(rec("DP=1;CH=x;UNDECLARED=a%2Cb;MQ=."), {"info_fields": ["DP"]}, ["DP"], "read")   # one character; '.'; a good escape
    both(rec("DP=1;CH=x"), {}, ["DP"], "refuse")                                      # Character with a builder: unsupported value type
    # ---- INFO: lists are walked only under a key with a builder
    both(rec("DP=1;LST=1,x,3"), {"info_fields": ["DP"]}, ["DP"], "read")
    both(rec("DP=1;LST=1,x,3"), {}, ["DP"], "refuse")
    both(rec("DP=1;LST=1,.,3"), {}, ["DP"], "read")
    # ---- `end`: entries in front of END are typed, those behind it are not; a single-base substitution asks nothing
    both(rec("MQ=--;END=20", ref="AC"), {}, ["chrom", "end"], "refuse")
    both(rec("END=20;MQ=--", ref="AC"), {}, ["chrom", "end"], "read")
    both(rec("MQ=--;END=20", ref="A"), {}, ["chrom", "end"], "read")
    both(rec("LST=1,x;END=20", ref="AC"), {}, ["chrom", "end"], "read")
    # ---- FORMAT: every value of a selected sample
    for s1 in ("1/1:6x", "1/1:6:--", "1/1:6:1.5:0,1,2:xy", "1/1:6:1.5:0,1,2:x:%FF"):
        fmt = "GT:GQ:XF:PL:FC:FT"
        s0 = "0/1:5:1.5:0,1,2:x:ok"
        both(rec("DP=1", fmt, s0, s1), {"format_fields": ["GT"]}, None, "refuse")
        both(rec("DP=1", fmt, s0, s1), {"format_fields": ["GT"]}, ["chrom", "DP"], "read")         # no FORMAT column
        both(rec("DP=1", fmt, s0, s1), {"format_fields": ["GT"], "samples": ["S0"]}, None, "read")  # the other sample
    both(rec("DP=1", "GT:GQ:XF:PL:FC:FT", "0/1:5:1.5:0,1,2:x:ok", "1/1:6:2:0,x,2:y:a%2Cb"), {"format_fields": ["GT", "FC"]}, None, "read")
    both(rec("DP=1", "GT:GQ:XF:PL:FC:FT", "0/1:5:1.5:0,1,2:x:ok", "1/1:6:2:0,x,2:y:ok"), {"format_fields": ["GT", "PL"]}, None, "refuse")
    both(rec("DP=1", "GT:GQ", "0/1:5", "1/x:6"), {"format_fields": ["GQ"]}, None, "read")           # a genotype nobody walks
    both(rec("DP=1", "GT:GQ", "0/1:5", "1/x:6"), {"format_fields": ["GT"]}, None, "refuse")
    both(rec("DP=1", "GT:GQ:FC", "0/1:5:x", "1/1:6:xy"), {"format_fields": ["GT", "FC"]}, None, "refuse")   # a selected Character
    # more FORMAT keys than the kernels' per-row map holds: the 17th value is typed too
    many = ":".join(["GT"] + [f"K{i}" for i in range(15)] + ["GQ"])
    vals = ":".join(["0/1"] + ["v"] * 15)
    both(rec("DP=1", many, vals + ":5", vals + ":6x"), {"format_fields": ["GT"]}, None, "refuse")
    both(rec("DP=1", many, vals + ":5", vals + ":6"), {"format_fields": ["GT"]}, None, "read")


def test_large_file_properties(pkg, tmp_path):
    """Size-independent properties on files too large for a value-by-value comparison (60 000 000 sites = BASELINE config 3 and
    200 000 x 1000 samples = config 4 when the box allows, else 4 M sites and 20 000 x 500): CRC32 + ISIZE of every member (a failure raises), every
    generated line comes back, `chrom = 'chr1'` returns exactly the generator's count, a second run gives the same
    totals, and the tabix plans of 8 and 16 partitions return the same number of rows in total; the multi-sample form
    returns lines x samples list cells and the same UDF checksum on a second run."""
    from conftest import scratch_dir, full_size_blocks
    # BASELINE configs 3 and 4 at their own size (60 M sites; 200 000 lines x 1000 samples) when the box has the scratch space
    # and the cores to write them in seconds, as the BAM twin does for config 2; BIOSCAN_TEST_LARGE_LINES overrides
    if os.environ.get("BIOSCAN_TEST_LARGE_LINES"):
        n_lines = int(os.environ["BIOSCAN_TEST_LARGE_LINES"])
    else:
        n_lines = full_size_blocks(60_000_000, 4_000_000, 45)
    full = n_lines >= 60_000_000
    base = scratch_dir(n_lines * 45)
    path = os.path.join(base, f"bioscan_large_{os.getpid()}.vcf.gz")
    spath = os.path.join(base, f"bioscan_large_{os.getpid()}_s.vcf.gz")
    try:
        meta = _synth(tmp_path, "sites", path, n_lines, 17, min(16, os.cpu_count() or 1))
        seq = pkg.VcfTableProvider(path, index_path="")
        plan = seq.scan(target_partitions=1)
        first = plan.execute_device(0, 8192)
        from conftest import report_size
        report_size("test_large_file_properties[vcf]", lines=n_lines, members=first["n_blocks"],
                    inflated_GB=round(first["inflated_bytes"] / 1e9, 2))
        assert first["n_rows"] == n_lines
        again = plan.execute_device(0, 8192)
        for k in ("n_rows", "n_blocks", "inflated_bytes", "arrow_bytes"):
            assert again[k] == first[k], k
        del seq, plan
        prov = pkg.VcfTableProvider(path)
        for target in (8, 16):
            p = prov.scan(target_partitions=target)
            assert sum(p.execute_device(i, 8192)["n_rows"] for i in range(p.num_partitions())) == n_lines, target
        p = prov.scan(filters=[("chrom", "=", "chr1")], target_partitions=8)
        assert sum(p.execute_device(i, 8192)["n_rows"] for i in range(p.num_partitions())) == meta["n_lines_chr1"]
        del prov, p

        lines, samples = (200_000, 1000) if full else (max(2000, n_lines // 200), 500)
        report_size("test_large_file_properties[vcf multi-sample]", lines=lines, samples=samples)
        _synth(tmp_path, "samples", spath, lines, samples, 19, min(16, os.cpu_count() or 1))
        ms = pkg.VcfTableProvider(spath, index_path="")
        mplan = ms.scan(target_partitions=1)
        a = mplan.execute_device_udf(0, "GQ", "list_avg")
        b = mplan.execute_device_udf(0, "GQ", "list_avg")
        assert a["scan"]["n_rows"] == lines and a["udf"]["n_rows"] == lines
        for k in ("n_rows", "n_elements", "count_a", "count_b", "sum"):
            assert a["udf"][k] == b["udf"][k], k
        assert a["udf"]["n_elements"] == lines * samples
        c = mplan.execute_device_udf(0, "DP", "list_gte", 0)
        assert c["udf"]["n_elements"] == lines * samples and c["udf"]["count_a"] + c["udf"]["count_b"] == lines * samples
    finally:
        for f in (path, path + ".tbi", spath, spath + ".tbi"):
            try:
                os.unlink(f)
            except OSError:
                pass


def test_differential_fuzz_of_whole_files(pkg, vo):
    """tools/fuzz_vcf_parity.py, a fixed number of files of two seeds: random headers (every Number / Type), 0..5 samples,
    missing values wherever the grammar allows them, escapes, lines spanning 150-byte BGZF members, one file in five with a
    malformed record -- schema, plan and rows against the oracle under random selections, projections and batch sizes; a file
    one side refuses the other refuses too -- including a value that does not type under a key the scan has no column for
    (noodles types every INFO entry and every value of a selected sample as it passes them)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_vcf_parity as F
    tot = dict(files=0, scans=0, indexed_scans=0, rows=0, refused_by_both=0)
    for seed in (5, 6):
        t, failures = F.run(pkg, seed=seed, max_files=60, verbose=False)
        assert not failures, failures[:3]
        for k in tot:
            tot[k] += t[k]
    assert tot["rows"] > 10000 and tot["refused_by_both"] > 5, tot
    from conftest import report_size
    report_size("test_differential_fuzz_of_whole_files", **tot)


@pytest.mark.parametrize("chunk", [1, 2, 7, 1000])
def test_chunked_stream_any_chunk_size(pkg, vo, tmp_path, chunk):
    """The chunk pipeline of a VCF stream (csrc/vcf_engine.cpp VcfChunkStream): members are inflated `chunk` at a time, the
    line cut by a chunk's end is carried to the next chunk, an indexed partition streams region after region and a batch
    that straddles chunks is concatenated on the host -- none of which may change a batch (the reference reads line by
    line in constant memory, bio-format-vcf/src/physical_exec.rs:912-1198).  Sites with every INFO column, indexed
    partitions and region filters on the reference's fixture, a limit, and a 1000-sample file whose lines span members."""
    path = os.path.join(G, "multi_chrom_large.vcf.gz")
    for tp in (1, 3, 8):
        assert _parity(pkg, vo, path, {}, target=tp, bs=77, chunk_members=chunk) == 10000
    assert _parity(pkg, vo, path, {}, names=["chrom", "start", "AF"], filters=[("chrom", "in", ["21", "22"])], target=4, bs=100,
                   chunk_members=chunk) == 10000
    n = _parity(pkg, vo, path, {}, names=["chrom", "id"], target=2, bs=13, chunk_members=chunk,
                filters=[("chrom", "=", "21"), ("start", ">=", 5009999), ("start", "<=", 5029999)])
    assert 0 < n < 5000
    assert _parity(pkg, vo, path, {}, names=["chrom"], filters=[("chrom", "=", "22")], limit=4100, bs=1000, chunk_members=chunk) == 4100
    # no index: one sequential partition
    dst = tmp_path / "noindex.vcf.gz"
    dst.write_bytes(open(path, "rb").read())
    assert _parity(pkg, vo, str(dst), {}, bs=333, chunk_members=chunk) == 10000
    assert _parity(pkg, vo, str(dst), {}, names=["qual", "filter", "DP"], limit=2500, bs=1000, chunk_members=chunk) == 2500
    # 1000 samples: a line is ~8 KB, the members 4 KB -- every line spans members, most chunks of one member hold no line end
    rng = random.Random(5 + chunk)
    rows = [[f"{rng.choice(['0/1', '1|1', './.', '0/0'])}:{rng.choice(['.', '7', '99', '30'])}:{rng.choice(['.', '12', '250'])}"
             for _ in range(1000)] for _ in range(25)]
    ms = tmp_path / "ms.vcf.gz"
    ms.write_bytes(bgzf_compress(_ms_vcf(rows).encode(), 4096))
    assert _parity(pkg, vo, str(ms), {"format_fields": ["GT", "GQ", "DP"]}, exact_batches=False, chunk_members=chunk) == 25
    assert _parity(pkg, vo, str(ms), {"format_fields": ["GT", "DP"], "samples": ["S3", "S999", "S0"]}, names=["start", "genotypes"],
                   exact_batches=False, chunk_members=chunk) == 25
