"""dev tool: differential fuzz of the BAM read path: random BAM files (tests/bam_build.py: record corners, aux fields, member
sizes down to 97 bytes) scanned by the GPU provider and by the oracle with random batch sizes, projections, coordinate systems,
CIGAR forms, chunk sizes of the stream pipeline and tag lists; every batch compared column by column."""
import os, random, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge
import bam_build as bb
import bam_oracle as oracle
pkg = ge._load_pkg(); pkg.load_library()


def cmp_batches(got, want, ctx):
    assert len(got) == len(want), (ctx, len(got), len(want))
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.num_rows == w.num_rows and g.schema.names == w.schema.names, (ctx, i)
        for name in w.schema.names:
            gc, wc = g.column(name), w.column(name)
            assert gc.type == wc.type, (ctx, name)
            if not gc.equals(wc):
                gl, wl = gc.to_pylist(), wc.to_pylist()
                bad = [k for k in range(len(wl)) if gl[k] != wl[k]][:3]
                raise AssertionError((ctx, i, name, [(k, gl[k], wl[k]) for k in bad]))


def rand_record(rng, k, n_ref, xb_sub):
    lseq = rng.choice([0, 1, 2, 15, 16, 17, 31, 33, 100, 150, 151, 255, 256, 257, 1000, rng.randrange(0, 9000) if k < 20 else 75])
    seq = "".join(rng.choice("=ACMGRSVTWYHKDBN" if rng.random() < 0.3 else "ACGT") for _ in range(lseq))
    qual = [rng.choice([0, 1, 40, 41, 93, 94, 95, 127, 200]) if rng.random() < 0.05 else rng.randrange(0, 94) for _ in range(lseq)]
    ncig = rng.choice([0, 1, 1, 2, 3, 4, 5, 8, 40])
    cigar = tuple((rng.choice([1, 9, 10, 99, 100, 12345, 268435455]), "MIDNSHP=X"[rng.randrange(9)]) for _ in range(ncig))
    name = rng.choice(["r", "*", "x" * rng.randrange(1, 255), "read/%d" % k, "a b", "q" * 15, "q" * 16, "q" * 17])
    refid = rng.randrange(-1, n_ref)
    nref = rng.randrange(-1, n_ref)
    aux = b""
    if rng.random() < 0.5:
        aux += bb.aux("NM", rng.choice("cCsSiI"), rng.randrange(0, 100))
    if rng.random() < 0.4:
        aux += bb.aux("MD", "Z", "".join(rng.choice("0123456789ACGT^") for _ in range(rng.randrange(0, 40))))
    if rng.random() < 0.3:
        aux += bb.aux("XB", "B" + xb_sub, [rng.randrange(0, 100) for _ in range(rng.randrange(0, 6))])
    if rng.random() < 0.3:
        aux += bb.aux("XA", "A", bytes([rng.randrange(33, 127)]))
    if rng.random() < 0.3:
        aux += bb.aux("XF", "f", rng.choice([0.0, 1.5, -2.25, 1e-40, 3.4e38]))
    return bb.record(name=name, refid=refid, pos=-1 if refid < 0 else rng.randrange(0, 900), mapq=rng.randrange(0, 256), flag=rng.randrange(0, 65536),
                     cigar=cigar, seq=seq, qual=qual, next_refid=nref, next_pos=-1 if nref < 0 else rng.randrange(0, 900),
                     tlen=rng.choice([0, 1, -1, 350, -350, 2 ** 31 - 1, -2 ** 31]), aux_bytes=aux)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    t0, cases, rows, errors = time.time(), 0, 0, 0
    tmp = tempfile.mkdtemp(prefix="fuzzbam")
    while time.time() - t0 < seconds:
        n_ref = rng.randrange(1, 12)
        refs = [("".join(rng.choice("abcdefghijklmnopqrstuvwxyz0123456789_") for _ in range(rng.choice([1, 2, 4, 5, 7, 8, 9, 16, 17, 30]))), 1000 + i) for i in range(n_ref)]
        if len(set(r[0] for r in refs)) != n_ref:
            continue
        nrec = rng.choice([0, 1, 2, 63, 64, 65, 255, 256, 257, 300, 1000])
        xb_sub = rng.choice("cCsSiIf")
        recs = [rand_record(rng, k, n_ref, xb_sub) for k in range(nrec)]
        member = rng.choice([97, 500, 4096, 60000])
        path = os.path.join(tmp, "f.bam")
        open(path, "wb").write(bb.bam(refs, recs, member=member))
        zero_based = rng.random() < 0.5
        binary = rng.random() < 0.3
        tags = rng.choice([None, ["NM"], ["NM", "MD", "XB", "XA", "XF"], ["XF", "NM"]])
        chunk = rng.choice([0, 1, 3, 64])
        prov = pkg.BamTableProvider(path, None, zero_based, tags, binary, index_path="", chunk_members=chunk) if chunk else pkg.BamTableProvider(path, None, zero_based, tags, binary, index_path="")
        orc = oracle.BamOracle(path, zero_based=zero_based, tag_fields=tags, index_path=None, binary_cigar=binary)
        names = orc.schema.names
        proj = rng.choice([None, None, [], [names.index("name")], [names.index("sequence"), names.index("quality_scores")],
                           [names.index("cigar"), names.index("end"), names.index("chrom")], sorted(rng.sample(range(len(names)), rng.randrange(1, len(names))))])
        bs = rng.choice([1, 7, 100, 256, 8192])
        ctx = (seed, cases, nrec, member, zero_based, binary, tags, chunk, proj, bs)
        gerr = oerr = None
        try:
            got = list(prov.scan(projection=proj).execute(0, bs))
        except pkg.BioscanError as e:
            gerr = str(e)
        try:
            _, want = orc.execute_sequential(proj, bs)
        except Exception as e:
            oerr = str(e)
        if gerr or oerr:
            assert gerr and oerr, (ctx, gerr, oerr)   # both sides refuse the file (a tag value that does not fit its column)
            errors += 1
        else:
            cmp_batches(got, want, ctx)
        cases += 1; rows += nrec
        if cases % 50 == 0: print(f"{cases} files, {rows} records", flush=True)
    print(f"OK: {cases} files, {rows} records compared, {errors} files refused by both sides")


if __name__ == "__main__":
    main()
