"""The write side of the BAM path: Arrow columns -> BAM records -> BGZF members on the GPU
(csrc/bam_write.hip, bam_writer.cpp), mirroring bio-format-bam/src/writer.rs + bio-format-core/src/sam_record_serializer.rs.
Parity bar: every member is BGZF that zlib inflates with the right CRC32 / ISIZE, and a file written from the reader's
batches reads back -- through the GPU reader AND the independent oracle -- as the same columns."""
import os
import random
import struct
import zlib

import pyarrow as pa
import pytest

import bam_build as bb
from test_gpu_bam_parity import _cmp_batches
from test_gpu_bam_edge_cases import REFS

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _members(data):
    """[(payload, member_bytes)] of a BGZF stream, each inflated with zlib and checked against its trailer."""
    out, o = [], 0
    while o < len(data):
        assert data[o:o + 4] == b"\x1f\x8b\x08\x04" and data[o + 12:o + 14] == b"BC", o
        bsize = struct.unpack_from("<H", data, o + 16)[0] + 1
        assert bsize <= 65536
        raw = zlib.decompressobj(-15)
        payload = raw.decompress(data[o + 18:o + bsize - 8]) + raw.flush()
        assert raw.eof and not raw.unused_data
        crc, isize = struct.unpack_from("<II", data, o + bsize - 8)
        assert isize == len(payload) and crc == zlib.crc32(payload), o
        out.append((payload, data[o:o + bsize]))
        o += bsize
    return out


def _corpus(rng):
    text = b"".join(b"@read%d/1\tACGTTGCA%s\t%s\n" % (i, b"ACGT"[i % 4:i % 4 + 1] * (i % 37), b"I" * (i % 53)) for i in range(20000))
    return [
        b"", b"A", b"AB", b"ABC", b"A" * 2, b"A" * 1000, b"abcabcabcabc" * 500, bytes(range(256)) * 300,
        bytes(rng.getrandbits(8) for _ in range(200000)),            # incompressible: stored blocks
        text, text[:65280], text[:65281], text[:65279], b"\0" * 300000,
        bytes(rng.choice(b"ACGT") for _ in range(150000)),
        b"".join(bytes([rng.randrange(200, 256)]) * rng.randrange(1, 300) for _ in range(2000)),   # long runs of 9-bit literals
    ]


def test_bgzf_deflate_members_are_valid_and_round_trip(pkg):
    rng = random.Random(4)
    for k, data in enumerate(_corpus(rng)):
        comp, ms = pkg.bgzf_deflate(data, add_eof=True)
        mem = _members(comp)
        assert mem[-1][0] == b"" and mem[-1][1] == comp[-28:]             # the BGZF EOF marker
        assert b"".join(p for p, _ in mem) == data, k
        assert all(len(p) <= 65280 for p, _ in mem)
        assert len(mem) == (len(data) + 65279) // 65280 + 1
        back, _ = pkg.bgzf_inflate(comp)                                   # K1 reads what the deflate kernel wrote
        assert back == data, k
        if len(data) >= 1000 and len(set(data)) <= 4:
            assert len(comp) < 0.7 * len(data), (k, len(comp), len(data))  # the match finder does find matches (random ACGT: ~0.63)
        if len(data) >= 1000 and len(set(data)) == 1:
            assert len(comp) < len(data) // 50 + 200, (k, len(comp), len(data))   # runs become 258-byte matches at distance 1


def test_compression_ratio_on_bam_like_data(pkg, golden):
    """Not a parity property, a sanity bound: the block's own Huffman codes + two match candidates per position + lazy
    evaluation on real BAM payload stay under 0.30 of the input (measured 0.278; zlib -6 reaches 0.2375 on the same bytes,
    the fixed code alone 0.58), and the members are dynamic-Huffman blocks (BTYPE = 10)."""
    data = open(os.path.join(golden, "multi_chrom_large.bam"), "rb").read()
    raw = b"".join(p for p, _ in _members(data))
    comp, _ = pkg.bgzf_deflate(raw)
    assert len(comp) < 0.30 * len(raw), (len(comp), len(raw))
    mem = _members(comp)
    assert all((m[18] >> 1) & 3 == 2 and m[18] & 1 for p, m in mem if len(p) > 1000)


def test_block_type_is_chosen_by_size(pkg):
    """Tiny members keep the fixed code (a dynamic header costs more than it saves), incompressible ones are stored, skewed
    alphabets get their own code -- and a literal/length alphabet that needs codes longer than 15 bits still decodes
    (length-limited codes)."""
    rng = random.Random(17)
    tiny, _ = pkg.bgzf_deflate(b"ACGT", add_eof=False)
    assert (tiny[18] >> 1) & 3 == 1
    rnd, _ = pkg.bgzf_deflate(bytes(rng.getrandbits(8) for _ in range(30000)), add_eof=False)
    assert (rnd[18] >> 1) & 3 == 0
    # Fibonacci-like frequencies: an unlimited Huffman tree would be ~25 levels deep
    fib = [1, 1]
    while len(fib) < 26:
        fib.append(fib[-1] + fib[-2])
    parts = []
    for k, f in enumerate(fib):
        parts.extend([k + 40] * min(f, 3000))
    rng.shuffle(parts)
    skew = bytes(parts)
    comp, _ = pkg.bgzf_deflate(skew, add_eof=False)
    assert (comp[18] >> 1) & 3 == 2
    assert b"".join(p for p, _ in _members(comp)) == skew
    back, _ = pkg.bgzf_inflate(comp + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    assert back == skew
    # one distinct byte / no matches at all (no distance code used) / exactly one distance code used
    for data in (b"Q" * 70000, bytes(range(256)) * 4, b"abcdefgh" * 3000):
        comp, _ = pkg.bgzf_deflate(data)
        assert b"".join(p for p, _ in _members(comp)) == data
        assert pkg.bgzf_inflate(comp)[0] == data


@pytest.mark.parametrize("fname", ["multi_chrom_large.bam", "nanopore_custom_tags.bam", "multi_chrom.bam"])
@pytest.mark.parametrize("zero_based", [True, False])
def test_written_file_reads_back_as_the_same_columns(pkg, oracle, tmp_path, fname, zero_based):
    src = os.path.join(G, fname)
    rd = pkg.BamTableProvider(src, None, zero_based, None, index_path="")
    so = oracle.BamOracle(src, zero_based=zero_based, index_path=None)
    batches = list(rd.scan().execute(0, 1000))
    out = str(tmp_path / "out.bam")
    w = pkg.BamWriter(out, so.hdr.text, so.hdr.ref_names, so.hdr.ref_lengths, zero_based)
    for b in batches:
        w.write_records(b)
    st = w.finish()
    assert st["n_records"] == sum(b.num_rows for b in batches)
    data = open(out, "rb").read()
    assert st["n_bytes"] == len(data)
    mem = _members(data)
    assert len(mem) == st["n_members"] + 1 and mem[-1][0] == b""
    # the GPU reader and the independent oracle both read the written file back as the original columns
    back = pkg.BamTableProvider(out, None, zero_based, None, index_path="")
    got = list(back.scan().execute(0, 1000))
    _cmp_batches(got, batches, (fname, "gpu read-back"))
    bo = oracle.BamOracle(out, zero_based=zero_based, index_path=None)
    _cmp_batches(bo.execute_sequential(None, 1000)[1], batches, (fname, "oracle read-back"))
    assert bo.hdr.ref_names == so.hdr.ref_names and bo.hdr.ref_lengths == so.hdr.ref_lengths and bo.hdr.text == so.hdr.text
    # the record bytes themselves: identical to the source's except for nothing -- bin, mapq, flags, names are all kept
    src_raw = b"".join(p for p, _ in _members(open(src, "rb").read()))
    new_raw = b"".join(p for p, _ in mem)
    so_first = so.hdr.first_record_offset if hasattr(so.hdr, "first_record_offset") else None
    if so_first is not None:
        a, b2 = src_raw[so_first:], new_raw[bo.hdr.first_record_offset:]
        # this scan projected no tag columns, so no aux fields are written: compare record by record up to the aux fields
        oa = ob = 0
        n = 0
        while oa < len(a):
            bs_a, bs_b = struct.unpack_from("<i", a, oa)[0], struct.unpack_from("<i", b2, ob)[0]
            ra, rb = a[oa + 4:oa + 4 + bs_a], b2[ob + 4:ob + 4 + bs_b]
            assert ra[:len(rb)] == rb, (fname, n)           # refID .. qual, bin included
            oa += 4 + bs_a
            ob += 4 + bs_b
            n += 1
        assert ob == len(b2) and n == st["n_records"]


def test_hand_built_records_round_trip(pkg, oracle, tmp_path):
    """Record corners: '*' and 254-byte names, every CIGAR op, empty and odd-length sequences, mates on '=' / other / none,
    negative template lengths, binary CIGAR input."""
    rng = random.Random(9)
    recs = []
    for k in range(200):
        lseq = rng.choice([0, 1, 2, 7, 150, 151, 1001])
        seq = "".join(rng.choice("=ACMGRSVTWYHKDBN") for _ in range(lseq))
        qual = [rng.randrange(0, 94) for _ in range(lseq)]
        ncig = rng.choice([0, 1, 3, 9])
        cigar = tuple((rng.choice([1, 5, 300, 268435455]), "MIDNSHP=X"[(k + j) % 9]) for j in range(ncig))
        refid = rng.choice([0, 1, 2, -1])
        recs.append(bb.record(name=rng.choice(["*", "r%d" % k, "x" * 254]), refid=refid, pos=-1 if refid < 0 else rng.randrange(0, 900),
                              mapq=rng.choice([0, 60, 255]), flag=rng.choice([0, 99, 147, 65535]), cigar=cigar, seq=seq, qual=qual,
                              next_refid=rng.choice([refid, 0, -1]), next_pos=rng.choice([-1, 5, 700]), tlen=rng.choice([0, -350, 2 ** 31 - 1, -2 ** 31])))
    src = str(tmp_path / "src.bam")
    open(src, "wb").write(bb.bam(REFS, recs))
    for binary in (False, True):
        rd = pkg.BamTableProvider(src, None, True, None, binary, index_path="")
        so = oracle.BamOracle(src, index_path=None)
        batches = list(rd.scan().execute(0, 64))
        out = str(tmp_path / ("out_%d.bam" % binary))
        w = pkg.BamWriter(out, so.hdr.text, so.hdr.ref_names, so.hdr.ref_lengths, True)
        for b in batches:
            w.write_records(b)
        assert w.finish()["n_records"] == 200
        _members(open(out, "rb").read())
        got = list(pkg.BamTableProvider(out, None, True, None, binary, index_path="").scan().execute(0, 64))
        _cmp_batches(got, batches, ("hand-built", binary))


def test_serializer_errors_are_the_references(pkg, tmp_path):
    names = ["name", "chrom", "start", "flags", "cigar", "mapping_quality", "mate_chrom", "mate_start", "sequence", "quality_scores",
             "template_length"]

    def batch(**over):
        cols = {"name": pa.array(["read1"]), "chrom": pa.array(["chr1"]), "start": pa.array([100], pa.uint32()),
                "flags": pa.array([0], pa.uint32()), "cigar": pa.array(["10M"]), "mapping_quality": pa.array([60], pa.uint32()),
                "mate_chrom": pa.array([None], pa.utf8()), "mate_start": pa.array([None], pa.uint32()),
                "sequence": pa.array(["ACGTACGTAC"]), "quality_scores": pa.array(["!!!!!!!!!!"]), "template_length": pa.array([0], pa.int32())}
        cols.update(over)
        return pa.RecordBatch.from_arrays([cols[n] for n in names if n in cols], names=[n for n in names if n in cols])

    def writer(k):
        return pkg.BamWriter(str(tmp_path / ("e%d.bam" % k)), "@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:249250621\n", ["chr1"], [249250621], True)
    w = writer(0)
    w.write_records(batch())                                             # serializer.rs:28-78: the basic record is fine
    assert w.finish()["n_records"] == 1
    with pytest.raises(pkg.BioscanError, match="does not fit into 16-bit SAM flags"):   # serializer.rs:80-124
        writer(1).write_records(batch(flags=pa.array([65536], pa.uint32())))
    with pytest.raises(pkg.BioscanError, match="CIGAR"):
        writer(2).write_records(batch(cigar=pa.array(["10Q"])))
    with pytest.raises(pkg.BioscanError, match="CIGAR"):
        writer(3).write_records(batch(cigar=pa.array(["M10"])))
    b = batch()
    with pytest.raises(pkg.BioscanError, match="Required column 'cigar' not found"):
        writer(4).write_records(b.drop_columns(["cigar"]))
    with pytest.raises(pkg.BioscanError, match="must be UInt32"):
        writer(5).write_records(batch(flags=pa.array([0], pa.int64())))
    # "*" forms (sam_record_serializer.rs:131-135, 225-237, 240-250): missing name, empty CIGAR, no sequence, no qualities
    w = writer(6)
    w.write_records(batch(name=pa.array(["*"]), cigar=pa.array(["*"]), sequence=pa.array(["*"]), quality_scores=pa.array(["*"])))
    w.finish()
    got = list(pkg.BamTableProvider(str(tmp_path / "e6.bam"), index_path="").scan().execute(0, 10))[0]
    assert got.column("name").to_pylist() == ["*"] and got.column("cigar").to_pylist() == [""] and got.column("sequence").to_pylist() == [""]


# ---- tag columns -> aux fields (build_tag_data, bio-format-core/src/sam_tag_io.rs:109-147, 206-656) -------------------------
def _aux_of(data):
    """per record: the bytes after the qualities"""
    raw = b"".join(p for p, _ in _members(data))
    l_text = struct.unpack_from("<i", raw, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]
    o += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", raw, o)[0]
        o += 4 + ln + 4
    out = []
    while o < len(raw):
        bs = struct.unpack_from("<i", raw, o)[0]
        lrn, ncig, lseq = raw[o + 12], struct.unpack_from("<H", raw, o + 16)[0], struct.unpack_from("<i", raw, o + 20)[0]
        a = o + 36 + lrn + 4 * ncig + (lseq + 1) // 2 + lseq
        out.append(raw[a:o + 4 + bs])
        o += 4 + bs
    return out


@pytest.mark.parametrize("fname,tags", [
    ("bam_with_tags.bam", ["NM", "MD", "MQ", "RG", "UQ", "XT", "XN", "OQ", "E2", "PG"]),
    ("nanopore_custom_tags.bam", ["NM", "AS", "ns", "pa", "de", "tp", "cm", "s1", "ms", "nn", "rl"]),
    ("10x_pbmc_tags.bam", ["CB", "CR", "CY", "UB", "UR", "UY", "NH", "HI", "nM", "AS", "RE", "xf"]),
    ("no_coor_only.bam", ["CB", "CR"]),
])
def test_tag_columns_are_written_as_aux_fields(pkg, oracle, tmp_path, fname, tags):
    src = os.path.join(G, fname)
    rd = pkg.BamTableProvider(src, None, True, tags, index_path="")
    so = oracle.BamOracle(src, zero_based=True, tag_fields=tags, index_path=None)
    batches = list(rd.scan().execute(0, 7))
    out = str(tmp_path / "out.bam")
    w = pkg.BamWriter(out, so.hdr.text, so.hdr.ref_names, so.hdr.ref_lengths, True)
    for b in batches:
        w.write_records(b)
    st = w.finish()
    data = open(out, "rb").read()
    # (1) the aux bytes of every record are what the oracle's restatement of build_tag_data gives for the same row
    aux = _aux_of(data)
    assert len(aux) == st["n_records"] == sum(b.num_rows for b in batches)
    k = 0
    any_aux = False
    for b in batches:
        for r in range(b.num_rows):
            want = oracle.build_tag_data(b, r)
            assert aux[k] == want, (fname, k, aux[k], want)
            any_aux = any_aux or bool(want)
            k += 1
    assert any_aux
    # (2) read back with the same tag list -- GPU reader and oracle -- the columns are the ones that were written
    got = list(pkg.BamTableProvider(out, None, True, tags, index_path="").scan().execute(0, 7))
    _cmp_batches(got, batches, (fname, "gpu read-back with tags"))
    bo = oracle.BamOracle(out, zero_based=True, tag_fields=tags, index_path=None)
    _cmp_batches(bo.execute_sequential(None, 7)[1], batches, (fname, "oracle read-back with tags"))


def _core(n):
    return {"name": pa.array(["r%d" % i for i in range(n)]), "chrom": pa.array(["chr1"] * n), "start": pa.array([100] * n, pa.uint32()),
            "flags": pa.array([0] * n, pa.uint32()), "cigar": pa.array(["4M"] * n), "mapping_quality": pa.array([60] * n, pa.uint32()),
            "mate_chrom": pa.array([None] * n, pa.utf8()), "mate_start": pa.array([None] * n, pa.uint32()),
            "sequence": pa.array(["ACGT"] * n), "quality_scores": pa.array(["!!!!"] * n), "template_length": pa.array([0] * n, pa.int32())}


def _tag_batch(n, tags):
    """tags: [(name, sam type spec or None, arrow array, has tag metadata)]"""
    cols = _core(n)
    fields = [pa.field(k, v.type) for k, v in cols.items()]
    arrays = list(cols.values())
    for name, spec, arr, marked in tags:
        md = {}
        if marked:
            md["bio.bam.tag.tag"] = name
            if spec is not None:
                md["bio.bam.tag.type"] = spec
        fields.append(pa.field(name, arr.type, metadata=md or None))
        arrays.append(arr)
    return pa.RecordBatch.from_arrays(arrays, schema=pa.schema(fields))


def test_every_tag_type_and_arrow_storage(pkg, oracle, tmp_path):
    """Every SAM type from every Arrow storage the reference accepts (extract_signed_int / extract_unsigned_int: 8..64-bit,
    Float32 / Float64, Utf8, List of those), NULLs skipped, columns without the tag metadata ignored, order = schema order."""
    n = 5
    tags = [
        ("XA", "A", pa.array(["a", "Z", None, "~", "0"]), True),
        ("XB", "A", pa.array([65, 0, 255, None, 97], pa.int64()), True),
        ("Xc", "c", pa.array([-128, 127, 0, None, -1], pa.int64()), True),
        ("XC", "C", pa.array([0, 255, 7, 8, None], pa.uint64()), True),
        ("Xs", "s", pa.array([-32768, 32767, None, 1, 2], pa.int16()), True),
        ("XS", "S", pa.array([0, 65535, 1, None, 3], pa.uint32()), True),
        ("Xi", "i", pa.array([-2 ** 31, 2 ** 31 - 1, 0, 5, None], pa.int32()), True),
        ("XI", "I", pa.array([0, 2 ** 32 - 1, None, 9, 1], pa.uint32()), True),
        ("Xu", "i", pa.array([0, 2 ** 31 - 1, 17, None, 1], pa.uint64()), True),
        ("Xf", "f", pa.array([1.5, -0.0, None, 3.4028234663852886e38, 1e-45], pa.float32()), True),
        ("Xg", "f", pa.array([0.1, None, -3.4028234663852886e38, 1e-50, 2.5], pa.float64()), True),
        ("XZ", "Z", pa.array(["", "hello world", None, "\u00e9", "x"]), True),
        ("XD", None, pa.array(["default is Z", None, "", "a", "b"]), True),
        ("XH", "H", pa.array(["", "1aFf", None, "00", "DEADBEEF"]), True),
        ("XQ", "Q", pa.array(["odd type char", None, "", "a", "b"]), True),          # unknown type character + string column: Z
        ("XR", "Q", pa.array([1, 2, 3, 4, 5], pa.int32()), True),                      # unknown type character + other column: skipped
        ("B1", "B:c", pa.array([[-128, 127], [], None, [0], [1, 2, 3]], pa.list_(pa.int64())), True),
        ("B2", "B:C", pa.array([[0, 255], None, [], [7], [1]], pa.list_(pa.uint8())), True),
        ("B3", "B:s", pa.array([[-32768], [32767], [], None, [0, 0]], pa.list_(pa.int32())), True),
        ("B4", "B:S", pa.array([[65535], [], [1, 2], [3], None], pa.list_(pa.uint16())), True),
        ("B5", "B:i", pa.array([[2 ** 31 - 1, -2 ** 31], [], None, [4], [5]], pa.list_(pa.int32())), True),
        ("B6", "B:I", pa.array([[2 ** 32 - 1], [0], [], None, [6]], pa.list_(pa.uint32())), True),
        ("B7", "B:f", pa.array([[1.0, -2.5], [], None, [0.1], [3.0]], pa.list_(pa.float32())), True),
        ("B8", "B:f", pa.array([[0.1, 1e-50], None, [], [2.0], [3.5]], pa.list_(pa.float64())), True),
        ("B9", "B", pa.array([[1, 2], [3], None, [], [4]], pa.list_(pa.uint16())), True),   # subtype from the Arrow element type
        ("no", "i", pa.array([1, 2, 3, 4, 5], pa.int32()), False),                     # no tag metadata: not a tag column
        ("LNG", "i", pa.array([1, 2, 3, 4, 5], pa.int32()), True),                     # name is not two bytes: skipped
    ]
    b = _tag_batch(n, tags)
    out = str(tmp_path / "t.bam")
    w = pkg.BamWriter(out, "@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:1000\n", ["chr1"], [1000], True)
    w.write_records(b)
    # a sliced batch: offsets of every column and of the list children are honoured
    w.write_records(b.slice(1, 3))
    w.finish()
    aux = _aux_of(open(out, "rb").read())
    want = [oracle.build_tag_data(b, r) for r in range(n)] + [oracle.build_tag_data(b.slice(1, 3), r) for r in range(3)]
    assert aux == want
    assert want[1] == want[5] and any(b"XQZ" in a for a in want) and not any(b"XR" in a or b"LNG" in a or b"noi" in a for a in want)


@pytest.mark.parametrize("spec,arr,msg", [
    ("c", pa.array([128], pa.int32()), "Integer value 128 does not fit SAM type 'c'"),
    ("C", pa.array([-1], pa.int32()), "Integer value -1 does not fit SAM type 'C'"),
    ("s", pa.array([40000], pa.uint16()), "Integer value 40000 does not fit SAM type 's'"),
    ("S", pa.array([65536], pa.int64()), "Integer value 65536 does not fit SAM type 'S'"),
    ("i", pa.array([2 ** 31], pa.int64()), "Integer value 2147483648 does not fit SAM type 'i'"),
    ("I", pa.array([2 ** 32], pa.uint64()), "Integer value 4294967296 does not fit SAM type 'I'"),
    ("i", pa.array(["7"]), "Tag value type mismatch for integer: Utf8"),
    ("f", pa.array([1e39], pa.float64()), "Float value 1000000000000000000000000000000000000000 does not fit SAM type 'f'"),
    ("f", pa.array([float("inf")], pa.float64()), "Float value inf does not fit SAM type 'f'"),
    ("f", pa.array([float("nan")], pa.float64()), "Float value NaN does not fit SAM type 'f'"),
    ("f", pa.array([1], pa.int32()), "Tag value type mismatch for float: Int32"),
    ("Z", pa.array([1], pa.int32()), "Tag value type mismatch for string: Int32"),
    ("H", pa.array(["abc"]), "Invalid SAM hex tag value 'ABC'"),
    ("H", pa.array(["zz"]), "Invalid SAM hex tag value 'ZZ'"),
    ("H", pa.array([1.0], pa.float32()), "Tag value type mismatch for hex string: Float32"),
    ("A", pa.array(["ab"]), "Character tags must be a single ASCII byte, got 'ab'"),
    ("A", pa.array([""]), "Character tags must be a single ASCII byte, got ''"),
    ("A", pa.array(["\u00e9"]), "Character tags must be a single ASCII byte, got '\u00e9'"),
    ("A", pa.array([256], pa.int32()), "Character tag value 256 does not fit into a single byte"),
    ("A", pa.array([-1], pa.int8()), "Character tag value -1 does not fit into a single byte"),
    ("A", pa.array([1.0], pa.float64()), "Tag value type mismatch for character: Float64"),
    ("B:c", pa.array([[1, 128]], pa.list_(pa.int32())), "Array element 128 does not fit SAM subtype 'c'"),
    ("B:I", pa.array([[0, -5]], pa.list_(pa.int64())), "Array element -5 does not fit SAM subtype 'I'"),
    ("B:f", pa.array([[1.0, 1e39]], pa.list_(pa.float64())), "Array element 1000000000000000000000000000000000000000 does not fit SAM subtype 'f'"),
    ("B:i", pa.array([[1, None]], pa.list_(pa.int32())), "SAM array tags cannot contain null elements"),
    ("B:i", pa.array([[1.0]], pa.list_(pa.float32())), "Unsupported array element type for SAM subtype 'i': Float32"),
    ("B:f", pa.array([[1]], pa.list_(pa.int32())), "Unsupported array element type for SAM subtype 'f': Int32"),
    ("B", pa.array([["x"]], pa.list_(pa.utf8())), "Unable to determine SAM array subtype for Arrow type Utf8"),
    ("B:i", pa.array([1], pa.int32()), "Tag value type mismatch for array: Int32"),
    ("ii", pa.array([1], pa.int32()), "Invalid SAM tag type metadata: Invalid SAM tag type 'ii': type must be a single character"),
    ("B:ii", pa.array([[1]], pa.list_(pa.int32())), "subtype must be a single character"),
    ("B:x", pa.array([[1]], pa.list_(pa.int32())), "Invalid SAM tag type metadata"),
    ("i:i", pa.array([1], pa.int32()), "expected 'TYPE' or 'B:SUBTYPE'"),
])
def test_tag_errors_are_the_references(pkg, oracle, tmp_path, spec, arr, msg):
    import re
    b = _tag_batch(1, [("XX", spec, arr, True)])
    with pytest.raises(oracle.TagWriteError, match=re.escape(msg)):
        oracle.build_tag_data(b, 0)
    w = pkg.BamWriter(str(tmp_path / "e.bam"), "@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:1000\n", ["chr1"], [1000], True)
    with pytest.raises(pkg.BioscanError, match=re.escape(msg)):
        w.write_records(b)
    # a NULL value is never converted: the same column with its value masked out writes a record without aux fields
    nb = _tag_batch(1, [("XX", spec if ":" not in spec[1:] or spec.startswith("B:") and len(spec) == 3 else "i",
                         pa.array([None], arr.type), True)])
    w2 = pkg.BamWriter(str(tmp_path / "n.bam"), "@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:1000\n", ["chr1"], [1000], True)
    w2.write_records(nb)
    w2.finish()
    assert _aux_of(open(str(tmp_path / "n.bam"), "rb").read()) == [b""]


def test_the_first_failing_row_is_reported(pkg, tmp_path):
    b = _tag_batch(4, [("XX", "c", pa.array([1, 300, 2, 400], pa.int32()), True), ("YY", "C", pa.array([1, 2, -7, 3], pa.int32()), True)])
    w = pkg.BamWriter(str(tmp_path / "f.bam"), "@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:1000\n", ["chr1"], [1000], True)
    with pytest.raises(pkg.BioscanError, match="Integer value 300 does not fit SAM type 'c'"):
        w.write_records(b)
