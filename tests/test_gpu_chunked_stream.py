"""The chunk pipeline of a BAM stream (bounded memory, csrc/engine.cpp BamExecState): a partition is inflated, framed
and extracted `chunk_members` BGZF members at a time, the record that straddles a chunk end is carried over, and a
RecordBatch that straddles two chunks is stitched on the host.  None of that may change a single batch: every chunk
size, down to one member per chunk, must reproduce the oracle's batches exactly -- the same parity bar as the
whole-partition tests (batches of exactly batch_size rows, bam/src/physical_exec.rs:545-565)."""
import os
import random

import pytest

import bam_build as bb
from test_gpu_bam_parity import _cmp_batches
from test_gpu_bam_edge_cases import REFS, _records

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("chunk", [1, 2, 3, 7, 1000])
@pytest.mark.parametrize("fname,tags", [("multi_chrom_large.bam", None), ("nanopore_custom_tags.bam", ["NM", "AS", "pa", "de", "tp"]),
                                        ("10x_pbmc_tags.bam", ["CB", "CR", "NH", "RG"])])
def test_sequential_scan_any_chunk_size(pkg, oracle, fname, tags, chunk):
    path = os.path.join(G, fname)
    prov = pkg.BamTableProvider(path, None, True, tags, index_path="", chunk_members=chunk)
    orc = oracle.BamOracle(path, zero_based=True, tag_fields=tags, index_path=None)
    for bs in (8192, 100, 7):
        got = list(prov.scan().execute(0, bs))
        _, want = orc.execute_sequential(None, bs)
        _cmp_batches(got, want, (fname, chunk, bs))
    got = list(prov.scan(projection=[]).execute(0, 33))
    _, want = orc.execute_sequential([], 33)
    assert [b.num_rows for b in got] == [b.num_rows for b in want]


@pytest.mark.parametrize("chunk", [1, 2, 5])
@pytest.mark.parametrize("target", [1, 3, 8])
def test_indexed_partitions_any_chunk_size(pkg, oracle, chunk, target):
    path = os.path.join(G, "multi_chrom_large.bam")
    prov = pkg.BamTableProvider(path, chunk_members=chunk)
    orc = oracle.BamOracle(path)
    for filters in ([], [("chrom", "in", ["chr2", "chrX"]), ("mapping_quality", ">=", 30)]):
        plan = prov.scan(filters=filters, target_partitions=target)
        parts, residual = orc.scan(filters=filters, target_partitions=target)
        assert plan.num_partitions() == len(parts)
        for p in range(plan.num_partitions()):
            got = list(plan.execute(p, 64))
            _, want = orc.execute_partition(parts[p].regions, None, residual, 64)
            _cmp_batches(got, want, ("indexed", chunk, target, p, filters))


@pytest.mark.parametrize("fname,tags", [("multi_chrom.bam", None), ("no_coor_only.bam", ["CB", "CR"]), ("bam_with_tags.bam", ["NM", "MD", "RG"])])
def test_unmapped_tails_and_no_coor_any_chunk_size(pkg, oracle, fname, tags):
    path = os.path.join(G, fname)
    orc = oracle.BamOracle(path, zero_based=True, tag_fields=tags)
    for chunk in (1, 2):
        prov = pkg.BamTableProvider(path, None, True, tags, chunk_members=chunk)
        for target in (1, 2, 4):
            plan = prov.scan(target_partitions=target)
            parts, residual = orc.scan(target_partitions=target)
            assert plan.num_partitions() == len(parts)
            for p in range(plan.num_partitions()):
                got = list(plan.execute(p, 10))
                _, want = orc.execute_partition(parts[p].regions, None, residual, 10)
                _cmp_batches(got, want, (fname, chunk, target, p))


@pytest.mark.parametrize("member,chunk", [(97, 1), (97, 5), (97, 64), (4096, 1), (4096, 3)])
def test_records_spanning_many_chunks(pkg, oracle, tmp_path, member, chunk):
    """97-byte members: a 5000-base record spans dozens of members, so with one member per chunk it is carried across
    dozens of chunks before it is complete; headers and block_size fields are cut by chunk ends."""
    rng = random.Random(member * 31 + chunk)
    recs = _records(rng, 60)
    path = str(tmp_path / "edge.bam")
    open(path, "wb").write(bb.bam(REFS, recs, member=member))
    tags = ["NM", "MD", "XA", "XB"]
    prov = pkg.BamTableProvider(path, None, True, tags, index_path="", chunk_members=chunk)
    orc = oracle.BamOracle(path, zero_based=True, tag_fields=tags, index_path=None)
    for bs in (8192, 7):
        got = list(prov.scan().execute(0, bs))
        _, want = orc.execute_sequential(None, bs)
        _cmp_batches(got, want, ("span", member, chunk, bs))


def test_truncated_last_record_is_an_error_in_every_chunking(pkg, tmp_path):
    rng = random.Random(5)
    payload_recs = _records(rng, 30)
    good = bb.bam(REFS, payload_recs, member=4096)
    # drop the last data member (keep the EOF marker): the record stream now ends inside a record
    eof = good[-28:]
    body = good[:-28]
    # find the start of the last member by re-framing
    o, starts = 0, []
    while o < len(body):
        starts.append(o)
        o += (body[o + 16] | (body[o + 17] << 8)) + 1
    cut = body[:starts[-1]] + eof
    path = str(tmp_path / "cut.bam")
    open(path, "wb").write(cut)
    for chunk in (1, 4, 0):
        prov = pkg.BamTableProvider(path, None, True, None, index_path="", chunk_members=chunk)
        with pytest.raises(pkg.BioscanError, match="record"):
            list(prov.scan().execute(0, 8192))


@pytest.mark.parametrize("index_unmapped", [True, False])
def test_unmapped_tails_do_not_decode_the_rest_of_the_file(pkg, oracle, tmp_path, index_unmapped):
    """An unmapped-tail scan starts at the reference's last chunk and ends where the reference changes -- its decode range
    must end there too (r03: it ran to the end of the file, so an indexed scan of a file with placed-unmapped reads on every
    reference inflated the file once per reference).  Four references, placed-unmapped reads behind the mapped ones of each,
    members of 2 000 bytes; with index_unmapped=False the index does not list those reads (their chunk is missing, so the
    tail really holds them), with True it does (the tail is empty: the last chunk ends where the next reference begins).
    Rows equal the oracle's per partition, and all partitions together inflate about one file's worth of members."""
    import struct
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bam_build as bb
    import fuzz_bam_indexed as F
    refs = [(f"chr{i + 1}", 2_000_000) for i in range(4)]
    recs, meta = [], []
    for refid in range(4):
        for k in range(600):
            pos = 1000 + k * 3000
            recs.append(bb.record(name=f"m{refid}_{k}", refid=refid, pos=pos, mapq=60, flag=99, cigar=((100, "M"),), seq="ACGT" * 25))
            meta.append((refid, pos, 100, 99, True))
        for k in range(3):
            pos = 1000 + 599 * 3000 + 50
            recs.append(bb.record(name=f"u{refid}_{k}", refid=refid, pos=pos, mapq=0, flag=69, cigar=(), seq="ACGT" * 25))
            meta.append((refid, pos, 0, 69, index_unmapped))
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs)
    tb = text.encode()
    h = b"BAM\1" + struct.pack("<i", len(tb)) + tb + struct.pack("<i", len(refs))
    for n, l in refs:
        nb = n.encode() + b"\0"
        h += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    payload = h + b"".join(recs)
    member = 2000
    data, coffs = F.bgzf_with_offsets(payload, member)
    full, u = [], len(h)
    for (refid, pos, span, flag, listed), r in zip(meta, recs):
        if listed:
            full.append((refid, pos, span, flag, u, u + len(r)))
        u += len(r)
    bai = bytearray(F.build_bai(4, full, lambda x: (coffs[x // member] << 16) | (x % member)))
    if not index_unmapped:
        # the pseudo-bin still says that unmapped reads exist (that is what makes the planner add the tail regions): patch
        # n_unmapped of every reference from 0 to 3
        o = 8
        for _ in range(4):
            n_bin = struct.unpack_from("<i", bai, o)[0]
            o += 4
            for _ in range(n_bin):
                b, n_chunk = struct.unpack_from("<Ii", bai, o)
                if b == 37450:
                    struct.pack_into("<Q", bai, o + 8 + 24, 3)
                o += 8 + 16 * n_chunk
            n_intv = struct.unpack_from("<i", bai, o)[0]
            o += 4 + 8 * n_intv
    path = tmp_path / "tails.bam"
    path.write_bytes(data)
    (tmp_path / "tails.bam.bai").write_bytes(bytes(bai))
    n_members = len(coffs) - 1
    prov = pkg.BamTableProvider(str(path))
    orc = oracle.BamOracle(str(path))
    for target in (1, 4):
        plan = prov.scan(target_partitions=target)
        parts, residual = orc.scan(target_partitions=target)
        assert plan.num_partitions() == len(parts)
        if target > 1:
            assert any(g.unmapped_tail for part in parts for g in part.regions)
        rows = inflated = 0
        for p in range(len(parts)):
            got = list(plan.execute(p, 500))
            _, want = orc.execute_partition(parts[p].regions, None, residual, 500)
            _cmp_batches(got, want, ("tails", index_unmapped, target, p))
            rows += sum(b.num_rows for b in got)
            inflated += plan.execute_device(p, 500)["n_blocks"]
        assert rows == (4 * 603 if index_unmapped else 4 * 600), rows   # (reads the index does not list and that have a position: no scan of the reference returns them)
        assert inflated <= n_members + 8 * len(parts) + 16, (inflated, n_members)


def test_an_error_stays_an_error_when_the_stream_is_polled_again(pkg, tmp_path):
    """after a failed chunk (here: a member whose CRC32 does not match) every later poll of the same stream
    fails again -- never the rows that were pending and then a clean end of stream, which would make a truncated partition
    look complete."""
    import ctypes as C
    rng = random.Random(11)
    good = bytearray(bb.bam(REFS, _records(rng, 400), member=2048))
    o, starts = 0, []
    while o < len(good):
        starts.append(o)
        o += (good[o + 16] | (good[o + 17] << 8)) + 1
    assert len(starts) > 12
    m = starts[len(starts) - 4]                       # a late data member: the chunks in front of it are fine
    end = m + (good[m + 16] | (good[m + 17] << 8)) + 1
    good[end - 8] ^= 0x5A                             # its CRC32 field
    path = str(tmp_path / "badcrc.bam")
    open(path, "wb").write(bytes(good))
    prov = pkg.BamTableProvider(path, None, True, None, index_path="", chunk_members=2)
    plan = prov.scan()
    lib = pkg.load_library()
    st = C.c_void_p()
    assert lib.bioscan_execute(plan._h, 0, 7, C.byref(st)) == 0
    try:
        import sys
        tp = sys.modules[pkg.__name__ + ".table_provider"]
        ok, failed = 0, 0
        for _ in range(10000):
            arr = tp._ArrowArray()
            has = C.c_int32()
            rc = lib.bioscan_next(st, C.addressof(arr), C.byref(has))
            if rc != 0:
                failed += 1
                if failed == 3:
                    break
                continue
            assert failed == 0, "a poll succeeded after the stream had failed"
            if not has.value:
                break
            ok += 1
            if arr.release:
                rel = C.CFUNCTYPE(None, C.c_void_p)(arr.release)
                rel(C.addressof(arr))
        assert ok > 0 and failed == 3, (ok, failed)
    finally:
        lib.bioscan_stream_close(st)
