"""GPU parity for the FASTQ path: HIP scan (C ABI) vs oracle/fastq_oracle.py, per partition and per
batch, on the reference's fixtures (sample.fastq.bgz + .gzi, sandbox example.fastq) plus edge cases;
and the reference's own properties (fastq/tests/parallel_read_test.rs)."""
import os
import shutil
import struct
import sys
import zlib

import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def fo():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fastq_oracle
    return fastq_oracle


def _run_gpu(pkg, path, target, projection=None, limit=None, bs=8192):
    prov = pkg.FastqTableProvider(path)
    plan = prov.scan(projection=projection, limit=limit, target_partitions=target)
    return prov, plan, [list(plan.execute(p, bs)) for p in range(plan.num_partitions())]


def _cmp(got, want, ctx):
    assert len(got) == len(want), (ctx, len(got), len(want))
    for g, w in zip(got, want):
        assert g.num_rows == w.num_rows, ctx
        assert g.schema.names == w.schema.names, ctx
        for n in w.schema.names:
            assert g.column(n).equals(w.column(n)), (ctx, n)


@pytest.mark.parametrize("target", [1, 2, 3, 4, 5, 6, 7, 8, 16])
def test_bgzf_gzi_partitions(pkg, fo, target):
    path = os.path.join(G, "sample.fastq.bgz")
    orc = fo.FastqOracle(path)
    strat, parts = orc.scan(target)
    prov, plan, got = _run_gpu(pkg, path, target, bs=300)
    assert prov.schema().equals(fo.SCHEMA)
    assert plan.num_partitions() == len(parts)
    total = 0
    for p, part in enumerate(parts):
        _, want = orc.execute(strat, part, batch_size=300)
        _cmp(got[p], want, (target, p))
        total += sum(b.num_rows for b in got[p])
    assert total == 2000                                         # parallel_read_test.rs:22-45


def test_no_duplicates_and_same_rows_1_vs_4(pkg):
    path = os.path.join(G, "sample.fastq.bgz")

    def rows(target):
        _, _, got = _run_gpu(pkg, path, target)
        out = []
        for part in got:
            for b in part:
                out += list(zip(b.column(0).to_pylist(), b.column(2).to_pylist(), b.column(3).to_pylist()))
        return out
    r1, r4 = rows(1), rows(4)
    assert len({r[0] for r in r4}) == 2000                        # parallel_read_test.rs:48-105
    assert sorted(r1) == sorted(r4)                               # parallel_read_test.rs:190-232


def test_no_gzi_is_sequential(pkg, fo, tmp_path):
    src = os.path.join(G, "sample.fastq.bgz")
    dst = str(tmp_path / "nogzi.fastq.bgz")
    shutil.copy(src, dst)
    prov, plan, got = _run_gpu(pkg, dst, 4)
    assert plan.num_partitions() == 1                             # parallel_read_test.rs:158-185
    _, want = fo.FastqOracle(dst).execute("sequential", None)
    _cmp(got[0], want, "nogzi")


@pytest.mark.parametrize("target", [1, 2, 3, 4, 8, 13])
def test_uncompressed_byte_ranges(pkg, fo, target):
    path = os.path.join(G, "example.fastq")
    orc = fo.FastqOracle(path)
    strat, parts = orc.scan(target)
    prov, plan, got = _run_gpu(pkg, path, target)
    assert plan.num_partitions() == len(parts)
    total = 0
    for p, part in enumerate(parts):
        _, want = orc.execute(strat, part)
        _cmp(got[p], want, (target, p))
        total += sum(b.num_rows for b in got[p])
    assert total == 200


def test_projection_limit_count(pkg, fo):
    path = os.path.join(G, "sample.fastq.bgz")
    orc = fo.FastqOracle(path)
    strat, parts = orc.scan(3)
    prov, plan, got = _run_gpu(pkg, path, 3, projection=[2, 0], limit=50)
    assert plan.display() == "FastqExec: projection=[sequence, name]"
    for p, part in enumerate(parts):
        _, want = orc.execute(strat, part, projection=[2, 0], limit=50)
        _cmp(got[p], want, p)
        assert sum(b.num_rows for b in got[p]) <= 50              # parallel_read_test.rs:134-155
    _, plan, got = _run_gpu(pkg, path, 4, projection=[])
    assert sum(b.num_rows for part in got for b in part) == 2000  # row_count_integration_test.rs


def _bgzf(chunks):
    out = b""
    for p in chunks:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(p) + c.flush()
        out += (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(body) + 8 - 1) + body +
                struct.pack("<II", zlib.crc32(p) & 0xFFFFFFFF, len(p)))
    return out + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def test_edge_cases(pkg, fo, tmp_path):
    """descriptions (space / tab / empty), CRLF, '@' leading a quality line, missing final newline, record
    boundaries exactly on block boundaries, records straddling several blocks, tiny blocks."""
    recs = []
    for i in range(400):
        name = f"r{i}"
        desc = ["", " desc text here", "\tx=1", " "][i % 4]
        seq = "ACGT" * (1 + i % 50)
        qual = ("@" if i % 3 == 0 else "I") + "#" * (len(seq) - 1)
        eol = "\r\n" if i % 7 == 0 else "\n"
        recs.append(f"@{name}{desc}{eol}{seq}{eol}+{eol}{qual}{eol}")
    text = "".join(recs).encode()
    text = text[:-1] if text.endswith(b"\n") else text          # no final newline
    cuts, o = [], 0
    k = 0
    while o < len(text):                                           # uneven blocks; some cuts land on record ends
        step = [len(recs[k % 400].encode()), 97, 4096, 5, 1500][k % 5]
        cuts.append(text[o:o + step])
        o += step
        k += 1
    data = _bgzf(cuts)
    path = str(tmp_path / "edge.fastq.bgz")
    open(path, "wb").write(data)
    # GZI from the block table
    ents, co, uo = [], 0, 0
    for c in cuts:
        bs = struct.unpack_from("<H", data, co + 16)[0] + 1
        co += bs
        uo += len(c)
        ents.append((co, uo))
    ents = ents[:-1]                                               # like bgzip -i: no entry for the EOF member
    open(path + ".gzi", "wb").write(struct.pack("<Q", len(ents)) + b"".join(struct.pack("<QQ", *e) for e in ents))
    plain = str(tmp_path / "edge.fastq")
    open(plain, "wb").write(text)
    for p_ in (path, plain):
        orc = fo.FastqOracle(p_)
        for target in (1, 2, 5, 9, 33):
            strat, parts = orc.scan(target)
            prov, plan, got = _run_gpu(pkg, p_, target, bs=64)
            assert plan.num_partitions() == len(parts)
            n = 0
            for p, part in enumerate(parts):
                _, want = orc.execute(strat, part, batch_size=64)
                _cmp(got[p], want, (p_, target, p))
                n += sum(b.num_rows for b in got[p])
            # The reference's resync only looks inside ONE buffered window (the rest of a BGZF block / an
            # 8 KiB BufReader window), so with blocks smaller than a record it can skip records at partition
            # starts; that behaviour is restated faithfully (GPU == oracle above).  Only the unsplit scan is
            # guaranteed complete.
            if target == 1:
                assert n == 400, (p_, target, n)


@pytest.mark.parametrize("tail", ["@r9 d\nACGT", "@r9 d\nACGT\n", "@r9 d\r\nACGT\r\n", "@r9\n"])
def test_record_cut_short_at_the_end_is_refused_by_both(pkg, fo, tmp_path, tail):
    """A file that ends inside a record, before its '+' line: noodles reads the '+' with read_exact (UnexpectedEof), so the
    reference's stream ends with an error -- on both sides here, plain and BGZF (tools/fuzz_fastq_parity.py seed 73 found the
    oracle accepting such a record with an empty quality line).  A record whose quality line is merely missing its newline, or is
    absent after a complete '+' line, is read."""
    good = "".join(f"@r{i} x\nACGTACGT\n+\nIIIIIIII\n" for i in range(5))
    for mode in ("plain", "bgzf"):
        text = (good + tail).encode()
        path = str(tmp_path / ("t.fastq" if mode == "plain" else "t.fastq.bgz"))
        open(path, "wb").write(text if mode == "plain" else _bgzf([text[:70], text[70:]]))
        orc = fo.FastqOracle(path)
        strat, parts = orc.scan(1)
        with pytest.raises(ValueError):
            orc.execute(strat, parts[0])
        with pytest.raises(pkg.BioscanError):
            _run_gpu(pkg, path, 1)
        ok = (good + "@r9 d\nACGT\n+\nIII").encode()          # the last line without its newline
        open(path, "wb").write(ok if mode == "plain" else _bgzf([ok[:70], ok[70:]]))
        orc = fo.FastqOracle(path)
        strat, parts = orc.scan(1)
        _, want = orc.execute(strat, parts[0])
        _, _, got = _run_gpu(pkg, path, 1)
        _cmp(got[0], want, (mode, "unterminated last line"))
        assert sum(b.num_rows for b in got[0]) == 6


def test_synthetic_fastq_partitions(pkg, fo, tmp_path):
    """medium scale: tools/synth_fastq (records straddle members), GPU == oracle per partition."""
    import json
    import subprocess
    exe = os.path.join(ROOT, "tools", "_build", "synth_fastq")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])
    path = str(tmp_path / "s.fastq.bgz")
    meta = json.loads(subprocess.check_output([exe, path, "200", "5", "8"]).decode())
    orc = fo.FastqOracle(path)
    for target in (1, 6):
        strat, parts = orc.scan(target)
        prov, plan, got = _run_gpu(pkg, path, target)
        n = 0
        for p, part in enumerate(parts):
            _, want = orc.execute(strat, part)
            _cmp(got[p], want, (target, p))
            n += sum(b.num_rows for b in got[p])
        assert n == meta["n_records"]


def test_random_line_lengths_and_line_ends(pkg, fo, tmp_path):
    """Randomised record shapes so that '\\n', '\\r', '@' and '+' land on every position of the newline kernels' 64-byte
    lane chunks and 16 KiB tiles (the index entries carry the bytes around each newline), plain and BGZF."""
    import random
    rng = random.Random(2025)
    recs = []
    for i in range(30000):
        name = "".join(rng.choice("abcXYZ09:._/") for _ in range(rng.randint(1, 40)))
        desc = rng.choice(["", " d", "\tq=1 z", " " + "x" * rng.randint(1, 70)])
        ln = rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 101, 150, rng.randint(1, 300)])
        seq = "".join(rng.choice("ACGTN") for _ in range(ln))
        qual = "".join(rng.choice("@+IJ#5?~!") for _ in range(ln))
        eol = rng.choice(["\n", "\n", "\r\n"])
        recs.append(f"@{name}{desc}{eol}{seq}{eol}+{eol}{qual}{eol}")
    text = "".join(recs).encode()
    plain = str(tmp_path / "rnd.fastq")
    open(plain, "wb").write(text)
    path = str(tmp_path / "rnd.fastq.bgz")
    cuts = [text[o:o + 60000] for o in range(0, len(text), 60000)]
    open(path, "wb").write(_bgzf(cuts))
    for p_ in (plain, path):
        orc = fo.FastqOracle(p_)
        strat, parts = orc.scan(1)
        prov, plan, got = _run_gpu(pkg, p_, 1, bs=8192)
        assert plan.num_partitions() == len(parts) == 1
        _, want = orc.execute(strat, parts[0], batch_size=8192)
        _cmp(got[0], want, (p_, "random"))
        assert sum(b.num_rows for b in got[0]) == 30000


def test_large_file_properties(pkg):
    """Size-independent properties on a file too large for a value-by-value comparison (32 768 members by default,
    BIOSCAN_TEST_LARGE_BLOCKS overrides): CRC32 + ISIZE of every member (a failure raises), every generated read comes
    back from the unsplit scan, a second run gives the same totals, and the GZI plan of 8 partitions neither loses nor
    duplicates a read (fastq/tests/parallel_read_test.rs: same rows for 1 and N partitions)."""
    import json
    import subprocess
    blocks = int(os.environ.get("BIOSCAN_TEST_LARGE_BLOCKS", "32768"))
    exe = os.path.join(ROOT, "tools", "_build", "synth_fastq")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])
    from conftest import scratch_dir
    base = scratch_dir(blocks * 27000)
    path = os.path.join(base, f"bioscan_large_{os.getpid()}.fastq.bgz")
    try:
        meta = json.loads(subprocess.check_output([exe, path, str(blocks), "13", str(min(16, os.cpu_count() or 1))]).decode())
        prov = pkg.FastqTableProvider(path)
        plan = prov.scan(target_partitions=1)
        first = plan.execute_device(0, 8192)
        from conftest import report_size
        report_size("test_large_file_properties[fastq]", members=meta["n_blocks"], reads=meta["n_records"],
                    inflated_GB=round(meta["inflated_bytes"] / 1e9, 2))
        assert first["n_rows"] == meta["n_records"]
        assert first["inflated_bytes"] == meta["inflated_bytes"]
        again = plan.execute_device(0, 8192)
        for k in ("n_rows", "n_blocks", "inflated_bytes", "arrow_bytes"):
            assert again[k] == first[k], k
        split = prov.scan(target_partitions=8)
        assert split.num_partitions() == 8
        parts = [split.execute_device(p, 8192) for p in range(8)]
        assert sum(s["n_rows"] for s in parts) == meta["n_records"]
        assert all(s["n_rows"] > 0 for s in parts)
    finally:
        for p in (path, path + ".gzi"):
            try:
                os.unlink(p)
            except OSError:
                pass


@pytest.mark.parametrize("chunk", [1, 2, 3, 7, 64, 1000])
def test_chunked_stream_any_chunk_size(pkg, fo, tmp_path, chunk):
    """The chunk pipeline of a BGZF FASTQ stream (csrc/engine.cpp FastqExecState): members are inflated `chunk` at a time,
    the record cut by a chunk's end is carried to the next chunk and batches continue across chunks -- none of which may
    change a batch.  GZI partitions of the reference's fixture (1 .. 8) and a file whose records straddle small members,
    every batch against the oracle (the reference reads record by record in constant memory,
    bio-format-fastq/src/physical_exec.rs:393-465)."""
    path = os.path.join(G, "sample.fastq.bgz")
    orc = fo.FastqOracle(path)
    for target in (1, 3, 8):
        strat, parts = orc.scan(target)
        prov = pkg.FastqTableProvider(path, chunk_members=chunk)
        plan = prov.scan(target_partitions=target)
        assert plan.num_partitions() == len(parts)
        for p, part in enumerate(parts):
            _, want = orc.execute(strat, part, batch_size=37)
            _cmp(list(plan.execute(p, 37)), want, (chunk, target, p))
    # records of random shapes over 700-byte members: a record spans several members, a chunk of one member may hold no
    # complete record at all
    import random
    rng = random.Random(77 + chunk)
    recs = []
    for i in range(1500):
        ln = rng.choice([1, 30, 101, 150, 700, rng.randint(1, 2000)])
        eol = rng.choice(["\n", "\r\n"])
        recs.append(f"@r{i} d{i % 7}{eol}{'ACGT' * (ln // 4 + 1)}{eol}+{eol}{'I' * (4 * (ln // 4 + 1))}{eol}")
    text = "".join(recs).encode()
    small = str(tmp_path / "small.fastq.bgz")
    open(small, "wb").write(_bgzf([text[o:o + 700] for o in range(0, len(text), 700)]))
    orc = fo.FastqOracle(small)
    strat, parts = orc.scan(1)
    _, want = orc.execute(strat, parts[0], batch_size=64)
    prov = pkg.FastqTableProvider(small, chunk_members=chunk)
    for limit in (None, 100):
        got = list(prov.scan(limit=limit).execute(0, 64))
        if limit is None:
            _cmp(got, want, (chunk, "small members"))
        else:
            assert sum(b.num_rows for b in got) == 100
            assert pa.Table.from_batches(got).equals(pa.Table.from_batches(want).slice(0, 100))


@pytest.mark.parametrize("chunk", [1, 2, 0])
def test_long_reads_across_many_members(pkg, fo, tmp_path, chunk):
    """Reads of 20 .. 90 kb (a nanopore-like file): a record spans several 64 KiB members, so a chunk of one member holds no
    complete record and the bytes carried to the next chunk cover several 16 KiB tiles of the text buffer -- the tiles whose
    newlines are counted partly by K2 (member bytes) and partly by the carry's own count (csrc/crc32.hip, fastq_kernels.hip)."""
    import random
    rng = random.Random(5 + chunk)
    recs = []
    for i in range(40):
        ln = rng.choice([20000, 33000, 65536, 90000, rng.randint(1, 50)])
        seq = "".join(rng.choice("ACGT") for _ in range(64)) * (ln // 64) + "A" * (ln % 64)
        eol = rng.choice(["\n", "\r\n"])
        recs.append(f"@long{i} ch={i % 5}{eol}{seq}{eol}+{eol}{'5' * len(seq)}{eol}")
    text = "".join(recs).encode()
    path = str(tmp_path / "long.fastq.bgz")
    open(path, "wb").write(_bgzf([text[o:o + 65280] for o in range(0, len(text), 65280)]))
    orc = fo.FastqOracle(path)
    strat, parts = orc.scan(1)
    _, want = orc.execute(strat, parts[0], batch_size=7)
    prov = pkg.FastqTableProvider(path, chunk_members=chunk)
    got = list(prov.scan().execute(0, 7))
    _cmp(got, want, ("long reads", chunk))
    assert sum(b.num_rows for b in got) == 40


def test_differential_fuzz_of_whole_files(pkg):
    """tools/fuzz_fastq_parity.py, a fixed number of files: random record shapes, line ends and descriptions, plain / BGZF with GZI
    / BGZF without, members of 60 .. 65 280 bytes, one file in eight with a broken record -- partition plans and every batch against
    the oracle under random target_partitions, pipeline chunk sizes (down to one member), batch sizes, projections and limits; a
    file one side refuses the other refuses too."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_fastq_parity as F
    t, failures = F.run(pkg, seed=11, max_files=80, verbose=False)
    assert not failures, failures[:3]
    assert t["rows"] > 20000 and t["scans"] > 80, t
    from conftest import report_size
    report_size("test_differential_fuzz_of_whole_files[fastq]", **t)
