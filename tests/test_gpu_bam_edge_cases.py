"""GPU parity on hand-built BAM records (tests/bam_build.py) for the corners of the per-record decode (SURVEY §8 a8,
bam/src/physical_exec.rs:412-540): empty sequence, odd / long sequences, no CIGAR, every CIGAR op, unmapped reads,
missing mates, read names of length 1 and 254, quality bytes >= 95 (two UTF-8 bytes each), records that span BGZF
members, negative template lengths, records with many aux fields, tiny and single-record files."""
import os
import random
import struct

import pytest

import bam_build as bb
from test_gpu_bam_parity import _cmp_batches

pytestmark = pytest.mark.gpu

REFS = [("chr1", 248956422), ("chrUn_KI270742v1", 186739), ("2", 1000)]


def _records(rng, n):
    recs = []
    for k in range(n):
        kind = k % 12
        seq_len = [0, 1, 2, 3, 7, 150, 151, 1000, 31, 32, 33, 5000][kind]
        seq = "".join(rng.choice("=ACMGRSVTWYHKDBN") for _ in range(seq_len))
        if kind in (5, 6):
            seq = "".join(rng.choice("ACGT") for _ in range(seq_len))
        qual = [rng.choice([0, 1, 40, 41, 93, 94, 95, 96, 127, 200, 222]) for _ in range(seq_len)]
        if kind == 4:
            qual = [rng.randrange(0, 94) for _ in range(seq_len)]
        ops = "MIDNSHP=X"
        ncig = [0, 1, 2, 9, 1, 3, 1, 40, 2, 0, 5, 1][kind]
        cigar = tuple((rng.choice([1, 2, 15, 100, 268435455 if kind == 3 else 77]), ops[(k + j) % 9]) for j in range(ncig))
        name = ["r", "x" * 254, "read/1", "a b", f"q{k}", "SRR1.1", "n" * 100, "é".encode().decode("latin-1")[:1] + "z", "*", "0", "READ", "r_r"][kind]
        refid = [0, 1, 2, -1, 0, 0, 1, 0, -1, 2, 0, 1][kind]
        pos = [-1 if refid < 0 else rng.randrange(0, 900), 0, 999, -1, 5, 100000, 186000, 7, 12, 0, 2 ** 29 - 1, 1][kind]
        nref = [-1, 0, 1, 2, -1, 0, 0, -1, 1, 2, 0, 1][(kind + 3) % 12]
        npos = -1 if nref < 0 else rng.randrange(0, 1000)
        aux = b""
        if kind in (5, 7, 10):
            aux = bb.aux("NM", "C", k % 250) + bb.aux("MD", "Z", "10A5^AC6") + bb.aux("XA", "A", b"Q") + bb.aux("XB", "BS", [1, 2, 65535])
        recs.append(bb.record(name=name, refid=refid, pos=pos, mapq=rng.choice([0, 1, 60, 254, 255]), flag=rng.choice([0, 4, 77, 141, 2048, 65535]),
                              cigar=cigar, seq=seq, qual=qual, next_refid=nref, next_pos=npos,
                              tlen=rng.choice([0, 1, -1, 2 ** 31 - 1, -2 ** 31, 350, -350]), aux_bytes=aux))
    return recs


@pytest.mark.parametrize("member", [60000, 4096, 97])
@pytest.mark.parametrize("zero_based", [True, False])
def test_record_corners(pkg, oracle, tmp_path, member, zero_based):
    rng = random.Random(member)
    recs = _records(rng, 240 if member > 100 else 48)
    path = str(tmp_path / "edge.bam")
    open(path, "wb").write(bb.bam(REFS, recs, member=member))   # small members: headers and records span BGZF blocks
    tags = ["NM", "MD", "XA", "XB"]
    prov = pkg.BamTableProvider(path, None, zero_based, tags, index_path="")
    orc = oracle.BamOracle(path, zero_based=zero_based, tag_fields=tags, index_path=None)
    assert prov.schema().equals(orc.schema, check_metadata=False)
    for bs in (8192, 7):
        got = list(prov.scan().execute(0, bs))
        _, want = orc.execute_sequential(None, bs)
        _cmp_batches(got, want, ("edge", member, zero_based, bs))
    # projections of single columns and COUNT(*)
    names = orc.schema.names
    for cols in ([names.index("quality_scores")], [names.index("cigar"), names.index("end")], []):
        got = list(prov.scan(projection=cols).execute(0, 8192))
        _, want = orc.execute_sequential(cols, 8192)
        _cmp_batches(got, want, ("edge-proj", cols))


def test_tiny_files(pkg, oracle, tmp_path):
    for n in (0, 1, 2):
        rng = random.Random(n)
        path = str(tmp_path / f"tiny{n}.bam")
        open(path, "wb").write(bb.bam(REFS, _records(rng, 12)[5:5 + n]))
        prov = pkg.BamTableProvider(path, None, True, None, index_path="")
        orc = oracle.BamOracle(path, zero_based=True, tag_fields=None, index_path=None)
        got = list(prov.scan().execute(0, 8192))
        _, want = orc.execute_sequential(None, 8192)
        _cmp_batches(got, want, ("tiny", n))
        assert sum(b.num_rows for b in got) == n


def _patched(rec: bytes, **fields) -> bytes:
    """record() bytes with raw header fields overwritten (block_size stays what it was)."""
    b = bytearray(rec)
    if "l_read_name" in fields:
        b[12] = fields["l_read_name"]
    if "n_cigar" in fields:
        struct.pack_into("<H", b, 16, fields["n_cigar"])
    if "l_seq" in fields:
        struct.pack_into("<i", b, 20, fields["l_seq"])
    return bytes(b)


@pytest.mark.parametrize("patch", [{"n_cigar": 65535}, {"l_seq": -5}, {"l_seq": 2 ** 31 - 1}, {"l_read_name": 0}, {"l_seq": 100000}])
def test_malformed_record_is_an_error(pkg, tmp_path, patch):
    """A CRC-valid member can carry a record whose variable-length fields do not fit in block_size: noodles fails
    such a record; the GPU path must report it instead of reading past the inflated buffer."""
    rng = random.Random(1)
    recs = _records(rng, 24)
    recs[13] = _patched(bb.record(name="bad", seq="ACGT" * 10, cigar=((40, "M"),)), **patch)
    path = str(tmp_path / "malformed.bam")
    open(path, "wb").write(bb.bam(REFS, recs))
    prov = pkg.BamTableProvider(path, None, True, None, index_path="")
    for proj in (None, [0], [3, 5]):  # (an empty projection interprets no record, like the reference's lazy records)
        with pytest.raises(pkg.BioscanError, match="invalid record"):
            list(prov.scan(projection=proj).execute(0, 8192))


def test_truncated_bgzf_header_is_an_error(pkg, tmp_path):
    """A file that ends inside a member's gzip extra field (page-multiple size, so nothing readable follows the mapping)."""
    rng = random.Random(2)
    good = bb.bam(REFS, _records(rng, 24))
    hdr = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 60000) + b"BC\x02\0"
    data = good + hdr
    data += b"\0" * ((-len(data)) % 4096)
    path = str(tmp_path / "trunc.bam")
    open(path, "wb").write(data)
    with pytest.raises(pkg.BioscanError, match="truncated block header|invalid block"):
        pkg.BamTableProvider(path, None, True, None, index_path="")


def test_reference_names_of_every_length(pkg, oracle, tmp_path):
    """chrom / mate_chrom are written from an 8-byte load of the name table (up to three stores) or, beyond 8 bytes, a byte
    loop: names of 1 .. 9, 15, 16, 17 and 40 bytes, the last one at the very end of the table."""
    lens = [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 40]
    refs = [("".join(chr(ord("a") + (i + j) % 26) for j in range(l)), 1000 + i) for i, l in enumerate(lens)]
    rng = random.Random(3)
    recs = []
    for k in range(600):
        r = k % len(refs)
        m = rng.randrange(-1, len(refs))
        ncig = k % 7                                                  # 0 .. 6 operations: both sides of the four-operation load
        cigar = tuple((rng.randrange(1, 400), "MIDNSHP=X"[(k + j) % 9]) for j in range(ncig))
        recs.append(bb.record(name="r%d" % k, refid=r, pos=rng.randrange(0, 900), cigar=cigar, seq="ACGT" * (k % 9), next_refid=m,
                              next_pos=-1 if m < 0 else rng.randrange(0, 900)))
    path = str(tmp_path / "refs.bam")
    open(path, "wb").write(bb.bam(refs, recs))
    for binary in (False, True):
        prov = pkg.BamTableProvider(path, None, True, None, binary, index_path="")
        orc = oracle.BamOracle(path, zero_based=True, index_path=None, binary_cigar=binary)
        for bs in (8192, 100):
            got = list(prov.scan().execute(0, bs))
            _, want = orc.execute_sequential(None, bs)
            _cmp_batches(got, want, ("ref names", binary, bs))


def test_csi_only_companion_plans_and_then_fails_like_the_reference(pkg, oracle, tmp_path):
    """`discover_bam_index` (bio-format-core/src/index_utils.rs:68-77) finds `<bam>.csi` when there is no `.bai`; the BAM
    provider then reads it with `bam::bai::fs::read` everywhere: `scan` gets unit size estimates (storage.rs:344-360) and no
    no-coor partition (:442-449), and every indexed partition fails at `IndexedBamReader::new` (storage.rs:286,
    physical_exec.rs:879-881: "Failed to open indexed BAM: ...").  r03 looked for `.bai` only and scanned sequentially."""
    import shutil
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multi_chrom.bam")
    bam = str(tmp_path / "only_csi.bam")
    shutil.copy(src, bam)
    with open(bam + ".csi", "wb") as f:
        f.write(b"CSI\x01" + bytes(64))
    prov = pkg.BamTableProvider(bam)
    orc = oracle.BamOracle(bam)
    assert orc.index_path == bam + ".csi" and orc.index_error is not None
    for target in (1, 2, 4):
        plan = prov.scan(target_partitions=target)
        parts, residual = orc.scan(target_partitions=target)
        assert plan.num_partitions() == len(parts), target
        for p in range(plan.num_partitions()):
            with pytest.raises(pkg.BioscanError, match="Failed to open indexed BAM"):
                list(plan.execute(p, 64))
            with pytest.raises(ValueError, match="Failed to open indexed BAM"):
                orc.execute_partition(parts[p].regions, None, residual, 64)
    # a scan without the index (index_path="") still reads the file
    seq = pkg.BamTableProvider(bam, index_path="")
    assert sum(b.num_rows for b in seq.scan().execute(0, 8192)) == 421
