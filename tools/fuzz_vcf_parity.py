"""dev tool: differential fuzz of the VCF read path.  Random VCF files -- INFO definitions of every Number / Type the schema
rules distinguish, FORMAT definitions, 0..5 samples, records with missing values in every position the grammar allows,
percent escapes, multi-allelic ALT, symbolic alleles, END, lines that span BGZF members of 150 .. 65 280 bytes -- are
scanned by the GPU provider and by oracle/vcf_oracle.py with random field selections, sample subsets, projections,
coordinate systems and batch sizes; schemas, plans and every partition's rows are compared (tests/test_gpu_vcf_parity.py:
_parity).  A file one side refuses must be refused by the other.  A share of the files carries one malformed record; six in ten
of the well-formed compressed ones get a tabix index built here and are also scanned through it (TBI size estimates, balanced
partitions, region queries, residual filters) with random region filters and target_partitions.
usage: fuzz_vcf_parity.py [seconds=60] [seed=1]"""
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import test_gpu_vcf_parity as T  # noqa: E402
import vcf_oracle as vo  # noqa: E402

INFO_POOL = [("DP", "1", "Integer"), ("AN", "1", "Integer"), ("AC", "A", "Integer"), ("AF", "A", "Float"), ("AD", "R", "Integer"),
             ("GL", "G", "Float"), ("MQ", "1", "Float"), ("DB", "0", "Flag"), ("SOMATIC", "0", "Flag"), ("GENE", "1", "String"),
             ("CSQ", ".", "String"), ("LST", ".", "Integer"), ("FL2", "2", "Float"), ("END", "1", "Integer"), ("NOTE", "1", "String"),
             ("IV3", "3", "Integer"), ("CH", "1", "Character"), ("CHL", ".", "Character")]
FORMAT_POOL = [("GT", "1", "String"), ("GQ", "1", "Integer"), ("DP", "1", "Integer"), ("AD", "R", "Integer"), ("PL", "G", "Integer"),
               ("FT", "1", "String"), ("XF", "1", "Float"), ("HQ", "2", "Integer"), ("GL", "G", "Float"), ("TAGS", ".", "String"),
               ("FC", "1", "Character")]
BASES = "ACGT"


def rint(rng):
    return rng.choice(["0", "1", "7", "42", "99", "250", "1000", "65535", "-1", "-17", "2147483647", "-2147483648", "+5", "007",
                       str(rng.randrange(-100000, 100000))])


def rfloat(rng):
    return rng.choice(["0", "0.5", "1e-3", "12.25", "0.1", "3.1e-42", "1E5", "-2.5", "100", ".5", "5.", "0.30000001", "1e38", "-0.0",
                       "%.*g" % (rng.randrange(1, 12), rng.random() * 10 ** rng.randrange(-8, 8)), str(rng.randrange(0, 1000) / 7.0)])


def rstr(rng):
    return rng.choice(["x", "abc", "BRCA1", "a_b-c", "p.Val600Glu", "x%3By", "50%25", "a%2Cb", "a|b|c", "longer_value_of_some_length_0123456789",
                       "".join(rng.choice("abcXYZ019_-|/()") for _ in range(rng.randrange(1, 30)))])


_MAY_BE_REFUSED = [False]   # make_vcf(malformed=True): values one side may refuse are allowed (such a file is never indexed)


def value(rng, typ):
    if typ == "Character":
        if _MAY_BE_REFUSED[0] and rng.random() < 0.03:
            return rng.choice(["a", "Z", "7", "%", "\u00e9", "ab"])
        return rng.choice("abcXYZ019")
    return rint(rng) if typ == "Integer" else rfloat(rng) if typ == "Float" else rstr(rng)


def values(rng, number, typ, n_alt, ploidy=2):
    if number == "1":
        n = 1
    elif number == "A":
        n = n_alt
    elif number == "R":
        n = n_alt + 1
    elif number == "G":
        n = (n_alt + 1) * (n_alt + 2) // 2
    elif number == ".":
        n = rng.randrange(1, 5)
    else:
        n = int(number)
    if number != "1" and rng.random() < 0.1:
        n = max(1, n + rng.choice([-1, 1]))          # the reference does not check a list's length against Number
    out = [("." if rng.random() < 0.12 else value(rng, typ)) for _ in range(max(1, n))]
    if rng.random() < 0.06:
        return "."
    return ",".join(out)


def gt(rng, n_alt):
    if rng.random() < 0.08:
        return rng.choice([".", "./.", ".|."])
    k = rng.choice([1, 2, 2, 2, 2, 3])
    al = [("." if rng.random() < 0.05 else str(rng.randrange(0, n_alt + 1))) for _ in range(k)]
    if rng.random() < 0.03:   # noodles parses an allele as an integer and the reference renders it again: "01" comes out as "1"
        al = [a if a == "." else rng.choice(["0", "00", "000"]) + a for a in al]
    sep = rng.choice(["/", "|"])
    s = sep.join(al) if rng.random() < 0.8 else "".join(a + rng.choice(["/", "|"]) for a in al)[:-1]
    return s


def make_vcf(rng, malformed):
    _MAY_BE_REFUSED[0] = malformed
    contigs = ["chr1", "chr2", "21", "X"][:rng.randrange(1, 5)]
    infos = rng.sample(INFO_POOL, rng.randrange(0, len(INFO_POOL) + 1))
    fmts = rng.sample(FORMAT_POOL, rng.randrange(1, len(FORMAT_POOL) + 1))
    n_samples = rng.choice([0, 0, 1, 2, 3, 5])
    samples = [f"S{i}" for i in range(n_samples)]
    hdr = ["##fileformat=VCFv4.3"]
    for c in contigs:
        hdr.append(f"##contig=<ID={c},length={rng.randrange(100000, 250000000)}>")
    if rng.random() < 0.5:
        hdr.append('##FILTER=<ID=q10,Description="Quality below 10">')
    for k, n, t in infos:
        hdr.append(f'##INFO=<ID={k},Number={n},Type={t},Description="{k} of the site">')
    if n_samples:
        for k, n, t in fmts:
            hdr.append(f'##FORMAT=<ID={k},Number={n},Type={t},Description="{k} of the sample">')
    cols = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"
    if n_samples:
        cols += "\tFORMAT\t" + "\t".join(samples)
    hdr.append(cols)
    n_rec = rng.choice([0, 1, 3, 40, 300, 2000])
    bad_at = rng.randrange(n_rec) if malformed and n_rec else -1
    lines = []
    for c in contigs:
        pos = rng.randrange(1, 1000)
        for r in range(n_rec // len(contigs) + 1):
            pos += rng.randrange(0, 500)
            ref = "".join(rng.choice(BASES) for _ in range(rng.choice([1, 1, 1, 2, 5, 30])))
            n_alt = rng.choice([0, 1, 1, 1, 2, 3])
            alts = []
            for _ in range(n_alt):
                alts.append(rng.choice(["".join(rng.choice(BASES) for _ in range(rng.choice([1, 1, 2, 8]))), "<DEL>", "*", "<INS:ME>", "N"]))
            alt = ",".join(alts) if alts else "."
            vid = rng.choice([".", ".", f"rs{rng.randrange(1, 10**9)}", f"rs{rng.randrange(1, 999)};id{rng.randrange(1, 99)}"])
            qual = rng.choice([".", "60", "29.5", "0", rfloat(rng).lstrip("-+") or "1"])
            filt = rng.choice([".", "PASS", "q10", "q10;s50"])
            items = []
            for k, n, t in infos:
                # (an INFO Character value is an error for a scan that has its column: only in files that may be refused)
                if rng.random() < (0.45 if t != "Character" else 0.9 if malformed else 1.0):
                    continue
                if t == "Flag":
                    items.append(k)
                elif k == "END":
                    items.append(f"END={pos + rng.randrange(0, 5000)}")
                else:
                    items.append(f"{k}={values(rng, n, t, n_alt)}")
            if rng.random() < 0.03:   # a key the header does not declare: String, Number=1
                items.append("UNDECL=" + rng.choice(["x", "a,b", "50%25", "1;", "%C3%A9"]).rstrip(";"))
            rng.shuffle(items)
            info = ";".join(items) if items else "."
            line = f"{c}\t{pos}\t{vid}\t{ref}\t{alt}\t{qual}\t{filt}\t{info}"
            if n_samples:
                keys = [f for f in fmts if rng.random() < 0.7] or [fmts[0]]
                if any(k == "GT" for k, _, _ in keys):                     # GT comes first when present
                    keys = [f for f in keys if f[0] == "GT"] + [f for f in keys if f[0] != "GT"]
                cells = []
                for _ in samples:
                    vals = []
                    for k, n, t in keys:
                        vals.append(gt(rng, n_alt) if k == "GT" else values(rng, n, t, n_alt))
                    cut = rng.choice([None, None, None, 1, 2])
                    if cut is not None:
                        vals = vals[:max(1, min(cut, len(vals)))]
                    cells.append("." if rng.random() < 0.04 else ":".join(vals))
                line += "\t" + ":".join(k for k, _, _ in keys) + "\t" + "\t".join(cells)
            if len(lines) == bad_at:
                kind = rng.randrange(9)
                f = line.split("\t")
                if kind == 0:
                    f[1] = rng.choice(["x", "", "-5", "12a", "99999999999", "4294967297", "18446744073709551616"])
                elif kind == 6 and infos:      # one entry that does not type, anywhere in INFO: an error for every scan with an INFO column
                    k, n, t = rng.choice(infos)
                    bad = "1x" if t == "Integer" else "--" if t == "Float" else "%FF" if t == "String" else "xy" if t == "Character" else "1"
                    if n != "1" and t != "Flag":
                        bad = "., " .strip() + bad if rng.random() < 0.5 else bad     # as a later element of a list
                    ents = [] if f[7] == "." else [e for e in f[7].split(";") if e.split("=")[0] != k]
                    ents.insert(rng.randrange(len(ents) + 1), f"{k}={bad}")
                    f[7] = ";".join(ents)
                elif kind == 7 and n_samples and len(f) > 9:   # one FORMAT value that does not type, in one sample
                    keys_here = f[8].split(":")
                    j = rng.randrange(len(keys_here))
                    decl = {k: (n, t) for k, n, t in fmts}.get(keys_here[j], ("1", "String"))
                    bad = "0/x" if keys_here[j] == "GT" else "5x" if decl[1] == "Integer" else "--" if decl[1] == "Float" else "xy" if decl[1] == "Character" else "%FF"
                    si = rng.randrange(9, len(f))
                    vals = f[si].split(":") if f[si] != "." else []
                    while len(vals) <= j:
                        vals.append(".")
                    vals[j] = bad
                    f[si] = ":".join(vals)
                elif kind == 8:
                    f[7] = (f[7] + ";" if f[7] != "." else "") + "UNDECL=%FF"
                elif kind == 1:
                    f = f[:rng.randrange(1, 8)]
                elif kind == 2 and infos:
                    k, n, t = infos[0]
                    f[7] = f"{k}=" + ("1x" if t == "Integer" else "--" if t == "Float" else "\x01") if t != "Flag" else f"{k}=1"
                elif kind == 3 and n_samples:
                    f[9] = "0/x" if "GT" in f[8].split(":")[:1] else "1x"
                elif kind == 4:
                    f[5] = rng.choice(["abc", "1..2", "--1"])
                else:
                    f[7] = "DP=1;DP=2" if any(k == "DP" for k, _, _ in infos) else f[7] + ";;"
                line = "\t".join(f)
            lines.append(line)
    return "\n".join(hdr + lines) + "\n", [k for k, _, _ in infos], [k for k, _, _ in fmts] if n_samples else [], samples


def _reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def bgzf_and_tbi(text: str, member: int):
    """-> (bgzf bytes, tbi bytes) for a position-sorted VCF text: BGZF members of `member` bytes and a tabix index built the way
    htslib builds one for VCF (names in order of appearance, span = END when INFO carries one, else POS .. POS + len(REF) - 1;
    bins with merged chunks, 16 kb linear index, the 37450 pseudo-bin)."""
    import struct
    import zlib
    raw = text.encode()
    out, coffs = bytearray(), []
    for ch in [raw[i:i + member] for i in range(0, len(raw), member)] + [b""]:
        coffs.append(len(out))
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = c.compress(ch) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(d) + 25) + d + struct.pack("<II", zlib.crc32(ch), len(ch))

    def voff(u):
        k, w = divmod(u, member)
        return (coffs[k] << 16) | w
    names, refs = [], []
    u = 0
    for line in text.split("\n")[:-1]:
        n = len(line.encode()) + 1
        if not line.startswith("#"):
            f = line.split("\t")
            if f[0] not in names:
                names.append(f[0])
                refs.append(dict(bins={}, lin={}, beg=None, end=None, n=0))
            r = refs[names.index(f[0])]
            beg = int(f[1]) - 1
            end = beg + max(1, len(f[3]))
            for ent in f[7].split(";"):
                if ent.startswith("END=") and ent[4:].isdigit() and int(ent[4:]) > beg:
                    end = int(ent[4:])
            v0, v1 = voff(u), voff(u + n)
            ch = r["bins"].setdefault(_reg2bin(beg, end), [])
            if ch and ch[-1][1] == v0:
                ch[-1] = (ch[-1][0], v1)
            else:
                ch.append((v0, v1))
            for w in range(beg >> 14, ((end - 1) >> 14) + 1):
                r["lin"].setdefault(w, v0)
            r["beg"] = v0 if r["beg"] is None else r["beg"]
            r["end"] = v1
            r["n"] += 1
        u += n
    nm = b"".join(x.encode() + b"\0" for x in names)
    t = bytearray(b"TBI\1" + struct.pack("<8i", len(names), 2, 1, 2, 0, ord("#"), 0, len(nm)) + nm)
    for r in refs:
        t += struct.pack("<i", len(r["bins"]) + 1)
        for b in sorted(r["bins"]):
            t += struct.pack("<Ii", b, len(r["bins"][b]))
            for c in r["bins"][b]:
                t += struct.pack("<QQ", *c)
        t += struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", r["beg"], r["end"], r["n"], 0)
        n_intv = max(r["lin"]) + 1
        t += struct.pack("<i", n_intv)
        last = 0
        for w in range(n_intv):
            last = r["lin"].get(w, last)
            t += struct.pack("<Q", last)
    t += struct.pack("<Q", 0)
    return bytes(out), T.bgzf_compress(bytes(t), 60000), names


def vcf_filters(rng, names):
    f = []
    k = rng.random()
    if k < 0.2:
        return f
    if k < 0.75:
        c = rng.choice(names + ["nope"])
        f.append(("chrom", "=", c))
        if rng.random() < 0.6:
            a = rng.randrange(0, 60000)
            b = a + rng.choice([1, 100, 5000, 40000, 10 ** 7])
            form = rng.random()
            if form < 0.4:
                f += [("start", ">=", a), ("start", "<=", b)]
            elif form < 0.6:
                f += [("start", ">=", a), ("end", "<=", b)]
            elif form < 0.8:
                f.append(("start", ">", a))
            else:
                f.append(("end", "<", b))
    else:
        f.append(("chrom", "in", [rng.choice(names + ["nope"]) for _ in range(rng.randrange(1, 4))]))
    if rng.random() < 0.2:
        f.append(("qual", ">=", 30.0))
    if rng.random() < 0.15:
        f.append(("id", "!=", "."))
    return f


def run(pkg, seconds=60.0, seed=1, max_files=None, verbose=True):
    """-> (totals dict, list of divergences)"""
    rng = random.Random(seed)
    t0 = time.time()
    n_files = n_rows = n_refused = n_scans = n_indexed = 0
    failures = []
    keep_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "fuzz_vcf_cases")

    def diverged(what, ctx, names, detail, path):
        failures.append((what, ctx, names, detail))
        print("DIVERGENCE:", what, ctx, names, detail[:400], flush=True)
        if len(failures) <= 6:
            os.makedirs(keep_dir, exist_ok=True)
            import shutil
            shutil.copy(path, os.path.join(keep_dir, f"seed{seed}_{os.path.basename(path)}"))
    with tempfile.TemporaryDirectory() as tmp:
        while (time.time() - t0 < seconds) if max_files is None else (n_files < max_files):
            malformed = rng.random() < 0.2
            text, info_keys, fmt_keys, samples = make_vcf(rng, malformed)
            gz = rng.random() < 0.7
            path = os.path.join(tmp, f"f{n_files}.vcf" + (".gz" if gz else ""))
            raw = text.encode()
            member = rng.choice([150, 211, 1000, 4096, 65280])
            tbi_names = None
            n_data = sum(1 for ln in text.split("\n") if ln and not ln.startswith("#"))
            if gz and not malformed and n_data and rng.random() < 0.6:
                bz, tbi, tbi_names = bgzf_and_tbi(text, member)
                with open(path, "wb") as f:
                    f.write(bz)
                with open(path + ".tbi", "wb") as f:
                    f.write(tbi)
            else:
                with open(path, "wb") as f:
                    f.write(T.bgzf_compress(raw, member) if gz else raw)
            n_files += 1
            if tbi_names is not None:
                # indexed scans: TBI size estimates, balanced partitions, region queries, residual filters
                for _ in range(rng.choice([2, 3])):
                    filters = vcf_filters(rng, tbi_names)
                    target = rng.choice([1, 2, 3, 5, 8])
                    kw = dict(zero_based=rng.random() < 0.5)
                    if rng.random() < 0.5:
                        kw["info_fields"] = rng.sample(info_keys, rng.randrange(0, len(info_keys) + 1))
                    if rng.random() < 0.5:
                        kw["format_fields"] = rng.sample(fmt_keys, rng.randrange(0, len(fmt_keys) + 1))
                    ctx = (seed, n_files - 1, path, kw, filters, target)
                    try:
                        n_rows += T._parity(pkg, vo, path, kw, filters=filters, target=target, bs=rng.choice([1, 100, 8192]),
                                            exact_batches=not (fmt_keys and len(samples) > 1))
                        n_indexed += 1
                    except AssertionError as e:
                        diverged("indexed scan differs", ctx, None, str(e), path)
                    except (vo.VcfError, ValueError, pkg.BioscanError) as e:
                        diverged("indexed scan raised", ctx, None, repr(e), path)
            for _ in range(rng.choice([1, 2, 3])):
                kw = dict(zero_based=rng.random() < 0.5)
                if rng.random() < 0.7:
                    kw["info_fields"] = rng.sample(info_keys, rng.randrange(0, len(info_keys) + 1))
                if rng.random() < 0.7:
                    kw["format_fields"] = rng.sample(fmt_keys, rng.randrange(0, len(fmt_keys) + 1))
                if samples and rng.random() < 0.4:
                    kw["samples"] = rng.sample(samples + ["nobody"], rng.randrange(1, len(samples) + 1))
                ctx = (seed, n_files - 1, path, kw)
                try:
                    o = vo.VcfOracle(path, **kw)
                    names = None
                    if rng.random() < 0.5:
                        names = rng.sample(o.schema.names, rng.randrange(0, len(o.schema.names) + 1))
                    want_err = None
                except (vo.VcfError, ValueError) as e:
                    want_err, names = e, None
                bs = rng.choice([1, 7, 100, 8192])
                if want_err is None:
                    try:
                        rows = T._parity(pkg, vo, path, kw, names=names, bs=bs, exact_batches=not (fmt_keys and len(samples) > 1))
                        n_rows += rows
                        n_scans += 1
                        continue
                    except (vo.VcfError, ValueError) as e:
                        want_err = e
                    except pkg.BioscanError as e:
                        # the GPU refused something: the oracle must refuse it too
                        try:
                            o = vo.VcfOracle(path, **kw)
                            plan = o.scan(projection=None if names is None else [o.schema.get_field_index(n) for n in names], filters=[], limit=None,
                                          target_partitions=1)
                            for p in range(o.num_partitions(plan)):
                                o.execute(plan, p, bs)
                        except (vo.VcfError, ValueError):
                            n_refused += 1
                            continue
                        diverged("GPU refused a file the oracle reads", ctx, names, str(e), path)
                        continue
                    except AssertionError as e:
                        diverged("rows differ", ctx, names, str(e), path)
                        continue
                    except Exception as e:   # (an exported batch pyarrow cannot read, ...)
                        diverged("scan raised " + type(e).__name__, ctx, names, repr(e), path)
                        continue
                # the oracle refused: so must the GPU
                try:
                    g = pkg.VcfTableProvider(path, kw.get("info_fields"), kw.get("format_fields"), None, kw.get("zero_based", True), kw.get("samples"))
                    sch = g.schema()
                    proj = None if names is None else [sch.get_field_index(n) for n in names]
                    plan = g.scan(projection=proj, filters=[], limit=None, target_partitions=1)
                    for p in range(plan.num_partitions()):
                        list(plan.execute(p, bs))
                except pkg.BioscanError:
                    n_refused += 1
                    continue
                diverged("the oracle refused a file the GPU reads", ctx, names, str(want_err), path)
            if verbose and n_files % 25 == 0:
                print(f"{n_files} files, {n_scans} scans, {n_rows} rows, {n_refused} refused by both sides", flush=True)
            for q in (path, path + ".tbi"):
                try:
                    os.unlink(q)
                except OSError:
                    pass
    totals = dict(files=n_files, scans=n_scans, indexed_scans=n_indexed, rows=n_rows, refused_by_both=n_refused)
    return totals, failures


def main():
    import __graft_entry__ as ge
    pkg = ge._load_pkg()
    pkg.load_library()
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t, failures = run(pkg, seconds, seed)
    print(f"{'OK' if not failures else 'FAILED'}: {t['files']} files, {t['scans']} scans + {t['indexed_scans']} indexed scans, {t['rows']} rows compared, "
          f"{t['refused_by_both']} scans refused by both sides, {len(failures)} divergences")
    if failures:
        sys.exit(1)


if __name__ == "__main__":
    main()
