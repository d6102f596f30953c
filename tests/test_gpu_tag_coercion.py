"""GPU parity for the tag-coercion corners (sam_tag_io.rs:658-1036) on BAM files written by
tests/bam_build.py: every aux type routed into every column type, Rust's f32 Display for a float landing in a Utf8
column, invalid UTF-8 -> NULL, range errors, type-mismatch errors.  The oracle is the checker."""
import random
import struct

import numpy as np
import pytest

import bam_build as bb
from test_gpu_bam_parity import _cmp_batches

pytestmark = pytest.mark.gpu

REFS = [("chr1", 100000), ("chr2", 50000)]


def _scan(pkg, oracle, path, tags, hints, bs=8192):
    prov = pkg.BamTableProvider(path, None, True, tags, infer_tag_types=False, tag_type_hints=hints, index_path="")
    orc = oracle.BamOracle(path, zero_based=True, tag_fields=tags, infer_tag_types=False, tag_type_hints=hints, index_path=None)
    assert prov.schema().equals(orc.schema, check_metadata=False)
    got = list(prov.scan().execute(0, bs))
    _, want = orc.execute_sequential(None, bs)
    return got, want


def _float_bits(rng, n):
    bits = [0x48A78AC4, 0xC9D5E5FA, 0x80000000, 0, 0x7F800000, 0xFF800000, 0x7FC00000, 1, 0x007FFFFF, 0x00800000, 0x7F7FFFFF,
            0x3F800000, 0x3DCCCCCD, 0x4B800000, 0x501502F9, 0x501502FA]
    bits += [struct.unpack("<I", struct.pack("<f", 10.0 ** k))[0] for k in range(-45, 39)]
    bits += [(e << 23) | f for e in range(0, 255, 3) for f in (0, 1, 0x7FFFFF)]
    bits += [rng.getrandbits(32) for _ in range(n)]
    return bits


def test_every_aux_type_into_a_utf8_column(pkg, oracle, tmp_path):
    rng = random.Random(11)
    recs = []
    vals = [("f", b) for b in _float_bits(rng, 3000)]
    vals += [("i", v) for v in (65, 0x20AC, 0x10FFFF, 0x110000, 0xD800, 0xDFFF, 0xE000, -1, -2147483648, 2147483647, 0, 127, 128, 0x7FF, 0x800, 0xFFFF, 0x10000)]
    vals += [("I", v) for v in (4294967295, 0x1F600, 3000000000)]
    vals += [("c", -128), ("C", 255), ("s", -32768), ("S", 65535), ("A", b"Q"), ("A", b"\xe9"), ("A", b"\x7f")]
    vals += [("Z", "hello"), ("Z", ""), ("Z", "zażółć"), ("Z", b"\xff\xfe"), ("Z", b"ab\xc3"), ("Z", b"\xed\xa0\x80"), ("Z", b"\xf4\x90\x80\x80"),
             ("Z", b"\xc0\xaf"), ("H", "1AE301"), ("Z", "x" * 300)]
    for k, (t, v) in enumerate(vals):
        a = bb.aux("XS", t, v)
        if k % 7 == 3:
            a = bb.aux("NM", "C", k % 200) + a  # another field in front
        if k % 11 == 10:
            a = b""  # absent -> NULL
        recs.append(bb.record(name=f"r{k}", pos=100 + k, aux_bytes=a))
    path = str(tmp_path / "utf8.bam")
    open(path, "wb").write(bb.bam(REFS, recs))
    got, want = _scan(pkg, oracle, path, ["XS"], ["XS:Z"])
    _cmp_batches(got, want, "utf8")
    col = [x for b in got for x in b.column("XS").to_pylist()]
    assert "343126.13" in col and "NaN" in col and "-inf" in col and "-0" in col and "16777216" in col


def test_integers_floats_and_arrays_into_typed_columns(pkg, oracle, tmp_path):
    rng = random.Random(12)
    recs = []
    for k in range(2000):
        a = b""
        t = rng.choice("cCsSiA")
        v = {"c": rng.randint(-128, 127), "C": rng.randint(0, 255), "s": rng.randint(-32768, 32767), "S": rng.randint(0, 65535),
             "i": rng.randint(-2 ** 31, 2 ** 31 - 1), "A": bytes([rng.randint(33, 126)])}[t]
        if k % 13:
            a += bb.aux("XI", t, v)
        t = rng.choice("CSIA")
        v = {"C": rng.randint(0, 255), "S": rng.randint(0, 65535), "I": rng.randint(0, 2 ** 32 - 1), "A": bytes([rng.randint(33, 126)])}[t]
        if k % 5:
            a += bb.aux("XU", t, v)
        if k % 3:
            a += bb.aux("XF", "f", rng.getrandbits(32) & ~(1 << 23))  # no NaNs: Arrow equality is by value
        st = rng.choice("cCsSiI")
        n = rng.choice([0, 1, 2, 7, 40])
        lo, hi = {"c": (-128, 127), "C": (0, 127), "s": (-128, 127), "S": (0, 127), "i": (-128, 127), "I": (0, 127)}[st]
        if k % 4:
            a += bb.aux("XB", "B" + st, [rng.randint(lo, hi) for _ in range(n)])
        if k % 6:
            a += bb.aux("XG", "Bf", [struct.unpack("<f", struct.pack("<I", rng.getrandbits(32) & 0x7F7FFFFF))[0] for _ in range(n)])
        st = rng.choice("CSI")
        if k % 2:
            a += bb.aux("XW", "B" + st, [rng.randint(0, {"C": 255, "S": 65535, "I": 2 ** 32 - 1}[st]) for _ in range(n)])
        recs.append(bb.record(name=f"q{k}", pos=10 + k, aux_bytes=a))
    path = str(tmp_path / "typed.bam")
    open(path, "wb").write(bb.bam(REFS, recs, member=20000))
    tags = ["XI", "XU", "XF", "XB", "XG", "XW"]
    hints = ["XI:i", "XU:I", "XF:f", "XB:B:c", "XG:B:f", "XW:B:I"]
    for bs in (8192, 333):
        got, want = _scan(pkg, oracle, path, tags, hints, bs)
        _cmp_batches(got, want, ("typed", bs))


@pytest.mark.parametrize("tag_aux,hint", [
    (bb.aux("XE", "I", 3000000000), "XE:i"),          # does not fit Int32
    (bb.aux("XE", "c", -1), "XE:I"),                  # negative into UInt32
    (bb.aux("XE", "f", 1.5), "XE:i"),                 # float into an integer builder
    (bb.aux("XE", "i", 7), "XE:f"),                   # integer into a Float32 builder
    (bb.aux("XE", "Z", "abc"), "XE:i"),               # string into an integer builder
    (bb.aux("XE", "Bs", [1, 300]), "XE:B:c"),         # array element out of Int8 range
    (bb.aux("XE", "Bf", [1.0]), "XE:B:i"),            # float array into an integer list
    (bb.aux("XE", "Bi", [1]), "XE:B:f"),              # integer array into a float list
    (bb.aux("XE", "Bc", [1]), "XE:Z"),                # array into a Utf8 column
    (bb.aux("XE", "i", 5), "XE:B:i"),                 # scalar into a list column
])
def test_coercion_errors_are_loud_on_both_sides(pkg, oracle, tmp_path, tag_aux, hint):
    recs = [bb.record(name="ok", pos=5, aux_bytes=b""), bb.record(name="bad", pos=6, aux_bytes=tag_aux)]
    path = str(tmp_path / "err.bam")
    open(path, "wb").write(bb.bam(REFS, recs))
    orc = oracle.BamOracle(path, zero_based=True, tag_fields=["XE"], infer_tag_types=False, tag_type_hints=[hint], index_path=None)
    with pytest.raises(Exception):
        orc.execute_sequential(None, 8192)
    prov = pkg.BamTableProvider(path, None, True, ["XE"], infer_tag_types=False, tag_type_hints=[hint], index_path="")
    with pytest.raises(RuntimeError):
        list(prov.scan().execute(0, 8192))
