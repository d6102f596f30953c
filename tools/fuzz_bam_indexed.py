"""dev tool: differential fuzz of the INDEXED BAM read path (BAI size estimates, balance_partitions, region queries, sub-region
dedup, residual filters, unmapped tails, the no-coordinate partition).  Random coordinate-sorted BAM files -- 1..5 references
of 20 kb .. 500 Mb, records clustered so that some 16 kb bins are crowded and most are empty, CIGARs whose reference span
crosses bin and linear-index boundaries, placed-unmapped reads, unplaced reads behind the last reference, BGZF members of 200
.. 60 000 bytes so that records span members -- get a BAI built here the way samtools builds one (bins by reg2bin with merged
chunks, 16 kb linear index, the 37450 pseudo-bin, n_no_coor) and are scanned by the GPU provider and by oracle/bam_oracle.py
with random target_partitions, filters (chrom = / IN, start / end bounds, mapping_quality, flags), projections, batch sizes,
coordinate systems and pipeline chunk sizes; plans and every partition's batches are compared.
usage: fuzz_bam_indexed.py [seconds=60] [seed=1]"""
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
import bam_build as bb  # noqa: E402
import bam_oracle as oracle  # noqa: E402

REF_SPAN_OPS = set("MDN=X")


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def bgzf_with_offsets(payload: bytes, member: int):
    """-> (file bytes, [compressed offset of member i], member size)"""
    out = bytearray()
    coffs = []
    chunks = [payload[i:i + member] for i in range(0, len(payload), member)] + [b""]
    for ch in chunks:
        coffs.append(len(out))
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = c.compress(ch) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(d) + 25)
        out += d + struct.pack("<II", zlib.crc32(ch), len(ch))
    return bytes(out), coffs


def build_bai(n_ref, recs, voff_of):
    """recs: [(refid, pos0, span, flag, u_start, u_end)] in file order; voff_of(u) -> virtual offset of uncompressed offset u.
    samtools' layout: per reference the bins with merged chunks, the linear index (smallest voffset of a record overlapping
    each 16 kb window, gaps filled from the left), the pseudo-bin 37450, then n_no_coor."""
    refs = [dict(bins={}, lin={}, beg=None, end=None, n_map=0, n_unmap=0) for _ in range(n_ref)]
    n_no_coor = 0
    for refid, pos0, span, flag, u0, u1 in recs:
        if refid < 0:
            n_no_coor += 1
            continue
        r = refs[refid]
        beg, end = pos0, pos0 + max(span, 1)
        v0, v1 = voff_of(u0), voff_of(u1)
        b = reg2bin(beg, end)
        ch = r["bins"].setdefault(b, [])
        if ch and ch[-1][1] == v0 and (ch[-1][1] >> 16) == (v0 >> 16):
            ch[-1] = (ch[-1][0], v1)          # adjacent records of one bin inside one member: one chunk
        else:
            ch.append((v0, v1))
        for w in range(beg >> 14, ((end - 1) >> 14) + 1):
            if w not in r["lin"]:
                r["lin"][w] = v0
        r["beg"] = v0 if r["beg"] is None else r["beg"]
        r["end"] = v1
        if flag & 4:
            r["n_unmap"] += 1
        else:
            r["n_map"] += 1
    out = bytearray(b"BAI\1" + struct.pack("<i", n_ref))
    for r in refs:
        n_bin = len(r["bins"]) + (1 if r["beg"] is not None else 0)
        out += struct.pack("<i", n_bin)
        for b in sorted(r["bins"]):
            out += struct.pack("<Ii", b, len(r["bins"][b]))
            for c in r["bins"][b]:
                out += struct.pack("<QQ", *c)
        if r["beg"] is not None:
            out += struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", r["beg"], r["end"], r["n_map"], r["n_unmap"])
        n_intv = (max(r["lin"]) + 1) if r["lin"] else 0
        out += struct.pack("<i", n_intv)
        last = 0
        for w in range(n_intv):
            last = r["lin"].get(w, last)
            out += struct.pack("<Q", last)
    out += struct.pack("<Q", n_no_coor)
    return bytes(out)


def make_bam(rng, want_records=False):
    n_ref = rng.randrange(1, 6)
    refs = [(f"chr{i + 1}" if rng.random() < 0.8 else f"scaffold_{i}", rng.choice([20000, 70000, 1 << 20, 50_000_000, 500_000_000])) for i in range(n_ref)]
    member = rng.choice([200, 1000, 4096, 20000, 60000])
    recs, meta = [], []
    n_per = rng.choice([0, 3, 40, 400])
    for refid, (_, length) in enumerate(refs):
        if rng.random() < 0.15:
            continue                                         # a reference without reads
        n = rng.randrange(0, n_per + 1)
        centres = [rng.randrange(0, length) for _ in range(rng.randrange(1, 5))]
        poss = sorted(min(length - 1, max(0, int(rng.gauss(rng.choice(centres), rng.choice([50, 3000, 200000]))))) for _ in range(n))
        for pos in poss:
            unmapped = rng.random() < 0.05
            l_seq = rng.choice([1, 10, 36, 150, 400])
            if unmapped:
                cigar, flag, span = (), 4 | (1 if rng.random() < 0.5 else 0), 0
            else:
                kind = rng.random()
                if kind < 0.7:
                    cigar = ((l_seq, "M"),)
                elif kind < 0.85:
                    a = rng.randrange(1, max(2, l_seq))
                    cigar = ((a, "S"), (max(1, l_seq - a), "M")) if l_seq > 1 else ((1, "M"),)
                else:
                    a = rng.randrange(1, max(2, l_seq))
                    cigar = ((a, "M"), (rng.choice([1, 100, 20000, 300000]), rng.choice("DN")), (max(1, l_seq - a), "M")) if l_seq > 1 else ((1, "M"),)
                span = sum(l for l, o in cigar if o in REF_SPAN_OPS)
                flag = rng.choice([0, 16, 99, 147, 83, 163, 1024 + 99])
                l_seq = sum(l for l, o in cigar if o in "MIS=X")
            seq = "".join(rng.choice("ACGT") for _ in range(l_seq))
            mate = rng.random() < 0.5
            recs.append(bb.record(name=f"r{len(recs)}", refid=refid, pos=pos, mapq=rng.choice([0, 20, 40, 60]), flag=flag, cigar=cigar, seq=seq,
                                  next_refid=refid if mate else -1, next_pos=pos + 200 if mate else -1, tlen=rng.randrange(-500, 500),
                                  aux_bytes=bb.aux("NM", "C", rng.randrange(0, 10)) if rng.random() < 0.5 else b""))
            meta.append((refid, pos, span, flag))
    for _ in range(rng.choice([0, 0, 2, 30])):                # unplaced reads behind the last reference
        l_seq = rng.choice([10, 150])
        recs.append(bb.record(name=f"u{len(recs)}", refid=-1, pos=-1, mapq=0, flag=4, cigar=(), seq="".join(rng.choice("ACGT") for _ in range(l_seq))))
        meta.append((-1, -1, 0, 4))
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs)
    tb = text.encode()
    h = b"BAM\1" + struct.pack("<i", len(tb)) + tb + struct.pack("<i", len(refs))
    for n, l in refs:
        nb = n.encode() + b"\0"
        h += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    payload = h + b"".join(recs)
    data, coffs = bgzf_with_offsets(payload, member)

    def voff_of(u):
        k, w = divmod(u, member)
        return (coffs[k] << 16) | w
    # one file in five has an index that does not list every read (an indexer that skips unmapped reads, or a stale index):
    # a record that lies between two chunks is then not part of any region's answer, although a decode of the span sees it
    omit = rng.random() < 0.2
    full, u = [], len(h)
    for (refid, pos, span, flag), r in zip(meta, recs):
        if not (omit and refid >= 0 and ((flag & 4) or rng.random() < 0.03)):
            full.append((refid, pos, span, flag, u, u + len(r)))
        u += len(r)
    if want_records:
        # (refid, pos0, span, flag, virtual offset of the record) of every record, listed by the index or not
        allr, u = [], len(h)
        for (refid, pos, span, flag), r in zip(meta, recs):
            allr.append((refid, pos, span, flag, voff_of(u)))
            u += len(r)
        return data, build_bai(len(refs), full, voff_of), refs, allr
    return data, build_bai(len(refs), full, voff_of), refs, len(recs)


def random_filters(rng, refs):
    names = [n for n, _ in refs]
    f = []
    k = rng.random()
    if k < 0.25:
        return f
    if k < 0.7:
        n, length = rng.choice(refs)
        f.append(("chrom", "=", n))
        if rng.random() < 0.6:
            a = rng.randrange(0, length)
            b = min(length, a + rng.choice([10, 1000, 20000, 400000, length]))
            form = rng.random()
            if form < 0.4:
                f += [("start", ">=", a), ("end", "<=", b)]
            elif form < 0.6:
                f.append(("start", "between", (a, b)))
            elif form < 0.8:
                f.append(("start", ">", a))
            else:
                f.append(("end", "<", b))
    else:
        f.append(("chrom", "in", [rng.choice(names + ["chrNone"]) for _ in range(rng.randrange(1, 4))]))
    if rng.random() < 0.3:
        f.append(("mapping_quality", ">=", rng.choice([1, 30, 60])))
    if rng.random() < 0.2:
        f.append(("flags", "!=", rng.choice([99, 4])))
    return f


def run(pkg, seconds=60.0, seed=1, max_files=None, verbose=True):
    from test_gpu_bam_parity import _cmp_batches
    rng = random.Random(seed)
    t0 = time.time()
    n_files = n_scans = n_rows = n_parts = n_refused = 0
    failures = []
    keep = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out", "fuzz_bam_indexed_cases")
    with tempfile.TemporaryDirectory() as tmp:
        while (time.time() - t0 < seconds) if max_files is None else (n_files < max_files):
            data, bai, refs, n_rec = make_bam(rng)
            path = os.path.join(tmp, f"f{n_files}.bam")
            with open(path, "wb") as f:
                f.write(data)
            with open(path + ".bai", "wb") as f:
                f.write(bai)
            n_files += 1
            for _ in range(rng.choice([2, 3, 4])):
                zero_based = rng.random() < 0.5
                filters = random_filters(rng, refs)
                target = rng.choice([1, 2, 3, 5, 8, 16])
                bs = rng.choice([1, 17, 1000, 8192])
                proj = None if rng.random() < 0.6 else sorted(rng.sample(range(12), rng.randrange(0, 6)))
                chunk = rng.choice([0, 0, 1, 2, 7])           # BGZF members per pipeline chunk of a stream (0 = default)
                ctx = (seed, n_files - 1, refs, filters, target, bs, proj, zero_based, chunk)
                def gpu_side():
                    prov = pkg.BamTableProvider(path, None, zero_based, None, chunk_members=chunk)
                    plan = prov.scan(projection=proj, filters=filters, target_partitions=target)
                    return plan, [list(plan.execute(p, bs)) for p in range(plan.num_partitions())]

                def oracle_side():
                    orc = oracle.BamOracle(path, zero_based=zero_based)
                    parts, residual = orc.scan(filters=filters, target_partitions=target)
                    return parts, [orc.execute_partition(part.regions, proj, residual, bs)[1] for part in parts]
                gerr = oerr = None
                try:
                    plan, got = gpu_side()
                except pkg.BioscanError as e:
                    gerr = e
                try:
                    parts, want = oracle_side()
                except (oracle.BamError if hasattr(oracle, "BamError") else ValueError, ValueError, KeyError) as e:
                    oerr = e
                try:
                    if gerr is not None or oerr is not None:
                        assert gerr is not None and oerr is not None, ("one side refuses", repr(gerr)[:300], repr(oerr)[:300])
                        n_refused += 1
                        continue
                    assert plan.num_partitions() == len(parts), ("partitions", plan.num_partitions(), len(parts))
                    for p in range(len(parts)):
                        _cmp_batches(got[p], want[p], (p, plan.partition_desc(p)))
                        n_rows += sum(b.num_rows for b in got[p])
                    n_parts += len(parts)
                    n_scans += 1
                except AssertionError as e:
                    failures.append((ctx, repr(e)[:600]))
                    print("DIVERGENCE:", ctx, repr(e)[:600], flush=True)
                    if len(failures) <= 5:
                        import shutil
                        os.makedirs(keep, exist_ok=True)
                        shutil.copy(path, os.path.join(keep, f"seed{seed}_f{n_files - 1}.bam"))
                        shutil.copy(path + ".bai", os.path.join(keep, f"seed{seed}_f{n_files - 1}.bam.bai"))
            if verbose and n_files % 25 == 0:
                print(f"{n_files} files, {n_scans} scans, {n_parts} partitions, {n_rows} rows", flush=True)
            os.unlink(path)
            os.unlink(path + ".bai")
    return dict(files=n_files, scans=n_scans, partitions=n_parts, rows=n_rows, refused_by_both=n_refused), failures


def main():
    import __graft_entry__ as ge
    pkg = ge._load_pkg()
    pkg.load_library()
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t, failures = run(pkg, seconds, seed)
    print(f"{'OK' if not failures else 'FAILED'}: {t['files']} files, {t['scans']} indexed scans, {t['partitions']} partitions, {t['rows']} rows compared, "
          f"{t['refused_by_both']} scans refused by both sides, {len(failures)} divergences")
    if failures:
        sys.exit(1)


if __name__ == "__main__":
    main()
