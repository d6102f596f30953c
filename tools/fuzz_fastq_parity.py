"""dev tool: differential fuzz of the FASTQ read path.  Random FASTQ files -- reads of 1 .. 3000 bases, names with and without
descriptions (space / tab separated, empty), LF and CRLF mixed, '@' and '+' at the start of quality lines, with and without a
final newline, now and then a record whose '+' line or name prefix is broken -- written plain, as BGZF with members of 60 ..
65 280 bytes (records span members; a member may hold no complete record) with a GZI index, or as BGZF without one, are scanned
by the GPU provider and by oracle/fastq_oracle.py with random target_partitions, pipeline chunk sizes, batch sizes,
projections and limits; the partition plans and every batch of every partition are compared.  A file one side refuses the
other must refuse.
usage: fuzz_fastq_parity.py [seconds=60] [seed=1]"""
import os
import random
import struct
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fastq_oracle as fo  # noqa: E402

EOFM = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bgzf(text: bytes, rng):
    """-> (bytes, gzi bytes): members of random sizes; the GZI lists every member but the first, like `bgzip -i`"""
    out, ents, o, co = [], [], 0, 0
    mode = rng.choice(["tiny", "small", "full", "mixed"])
    while o < len(text):
        step = {"tiny": rng.randint(60, 300), "small": rng.randint(300, 5000), "full": 65280,
                "mixed": rng.choice([61, 97, 700, 4096, 65280])}[mode]
        p = text[o:o + step]
        c = zlib.compressobj(rng.choice([1, 6, 9]), zlib.DEFLATED, -15)
        body = c.compress(p) + c.flush()
        m = (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(body) + 8 - 1) + body +
             struct.pack("<II", zlib.crc32(p) & 0xFFFFFFFF, len(p)))
        out.append(m)
        o += len(p)
        co += len(m)
        ents.append((co, o))
    ents = ents[:-1]
    gzi = struct.pack("<Q", len(ents)) + b"".join(struct.pack("<QQ", *e) for e in ents)
    return b"".join(out) + EOFM, gzi


def make_fastq(rng, malformed):
    n = rng.choice([0, 1, 2, 40, 400, 3000])
    shape = rng.choice(["short", "illumina", "long", "mixed"])
    bad_at = rng.randrange(n) if malformed and n else -1
    recs = []
    for i in range(n):
        name = "".join(rng.choice("abcXYZ09:._/#") for _ in range(rng.randint(1, 40)))
        desc = rng.choice(["", "", " 1:N:0:ATCACG", "\tq=1 z", " ", " " + "x" * rng.randint(1, 90)])
        ln = {"short": rng.randint(1, 40), "illumina": rng.choice([101, 150, 151]), "long": rng.randint(500, 3000),
              "mixed": rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 101, 150, rng.randint(1, 1200)])}[shape]
        seq = "".join(rng.choice("ACGTN") for _ in range(ln))
        qual = "".join(rng.choice("@+IJ#5?~!F:,") for _ in range(ln))
        eol = rng.choice(["\n", "\n", "\n", "\r\n"])
        plus = "+" if rng.random() < 0.8 else "+" + name
        rec = f"@{name}{desc}{eol}{seq}{eol}{plus}{eol}{qual}{eol}"
        if i == bad_at:
            kind = rng.randrange(4)
            if kind == 0:
                rec = f"{name}{desc}{eol}{seq}{eol}+{eol}{qual}{eol}"          # no '@'
            elif kind == 1:
                rec = f"@{name}{desc}{eol}{seq}{eol}-{eol}{qual}{eol}"         # no '+'
            elif kind == 2:
                rec = f"@{name}{desc}{eol}{seq}{eol}"                          # half a record
            else:
                rec = f"@{name}{desc}{eol}{seq}{eol}+{eol}{qual[:max(0, ln - 1)]}{eol}" if ln > 1 else rec   # short quality line
        recs.append(rec)
    text = "".join(recs).encode()
    if text.endswith(b"\n") and rng.random() < 0.3:
        text = text[:-2] if text.endswith(b"\r\n") else text[:-1]
    return text, n


def _cmp(got, want, ctx):
    assert len(got) == len(want), (ctx, "batches", len(got), len(want))
    for k, (g, w) in enumerate(zip(got, want)):
        assert g.num_rows == w.num_rows, (ctx, k, g.num_rows, w.num_rows)
        assert g.schema.names == w.schema.names, (ctx, k)
        for n in w.schema.names:
            assert g.column(n).equals(w.column(n)), (ctx, k, n)


def run(pkg, seconds=60.0, seed=1, max_files=None, verbose=True):
    rng = random.Random(seed)
    t0 = time.time()
    n_files = n_scans = n_parts = n_rows = n_refused = 0
    failures = []
    keep_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "fuzz_fastq_cases")

    def diverged(what, ctx, detail, paths):
        failures.append((what, ctx, detail))
        print("DIVERGENCE:", what, ctx, detail[:400], flush=True)
        if len(failures) <= 6:
            os.makedirs(keep_dir, exist_ok=True)
            import shutil
            for q in paths:
                if os.path.exists(q):
                    shutil.copy(q, os.path.join(keep_dir, f"seed{seed}_{os.path.basename(q)}"))
    with tempfile.TemporaryDirectory() as tmp:
        while (time.time() - t0 < seconds) if max_files is None else (n_files < max_files):
            malformed = rng.random() < 0.12
            text, n_rec = make_fastq(rng, malformed)
            form = rng.choice(["plain", "bgzf+gzi", "bgzf+gzi", "bgzf+gzi", "bgzf"])
            path = os.path.join(tmp, f"f{n_files}.fastq" + ("" if form == "plain" else ".bgz"))
            paths = [path]
            if form == "plain":
                open(path, "wb").write(text)
            else:
                data, gzi = bgzf(text, rng)
                open(path, "wb").write(data)
                if form == "bgzf+gzi":
                    open(path + ".gzi", "wb").write(gzi)
                    paths.append(path + ".gzi")
            n_files += 1
            for _ in range(rng.choice([1, 2, 3])):
                target = rng.choice([1, 1, 2, 3, 5, 8, 16])
                chunk = rng.choice([0, 0, 1, 2, 5, 64])
                bs = rng.choice([1, 7, 64, 300, 8192])
                proj = rng.choice([None, None, [0], [2, 3], [3, 1, 0], []])
                limit = rng.choice([None, None, None, 0, 1, 57])
                ctx = (seed, n_files - 1, form, len(text), n_rec, dict(target=target, chunk=chunk, bs=bs, proj=proj, limit=limit))
                try:
                    orc = fo.FastqOracle(path)
                    strat, parts = orc.scan(target)
                    want = [orc.execute(strat, part, projection=proj, limit=limit, batch_size=bs)[1] for part in parts]
                    want_err = None
                except (ValueError, fo.FastqError if hasattr(fo, "FastqError") else ValueError) as e:
                    want_err = e
                try:
                    prov = pkg.FastqTableProvider(path, chunk_members=chunk)
                    plan = prov.scan(projection=proj, limit=limit, target_partitions=target)
                    got = [list(plan.execute(p, bs)) for p in range(plan.num_partitions())]
                    got_err = None
                except pkg.BioscanError as e:
                    got_err = e
                if want_err is not None or got_err is not None:
                    if want_err is not None and got_err is not None:
                        n_refused += 1
                    else:
                        diverged("only one side refuses", ctx, f"oracle: {want_err!r}; gpu: {got_err!r}", paths)
                    continue
                try:
                    assert len(got) == len(want), ("partitions", len(got), len(want))
                    for p in range(len(want)):
                        _cmp(got[p], want[p], p)
                        n_rows += sum(b.num_rows for b in got[p])
                    n_parts += len(want)
                    n_scans += 1
                except AssertionError as e:
                    diverged("batches differ", ctx, str(e), paths)
                except Exception as e:
                    diverged("comparison raised " + type(e).__name__, ctx, repr(e), paths)
            if verbose and n_files % 50 == 0:
                print(f"{n_files} files, {n_scans} scans, {n_rows} rows", flush=True)
            for q in paths:
                try:
                    os.unlink(q)
                except OSError:
                    pass
    return dict(files=n_files, scans=n_scans, partitions=n_parts, rows=n_rows, refused_by_both=n_refused), failures


def main():
    import __graft_entry__ as ge
    pkg = ge._load_pkg()
    pkg.load_library()
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t, failures = run(pkg, seconds, seed)
    print(f"{'OK' if not failures else 'FAILED'}: {t['files']} files, {t['scans']} scans, {t['partitions']} partitions, {t['rows']} rows compared, "
          f"{t['refused_by_both']} scans refused by both sides, {len(failures)} divergences")
    if failures:
        sys.exit(1)


if __name__ == "__main__":
    main()
