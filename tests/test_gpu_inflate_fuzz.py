"""GPU: K1 parity against zlib on synthetic BGZF members that exercise every DEFLATE block type
and the edge cases of the format (stored / fixed / dynamic blocks, several blocks per member, empty
members, 1-byte members, 64 KiB members, long runs, incompressible data, long-distance matches)."""
import os
import random
import struct
import zlib

import pytest

pytestmark = pytest.mark.gpu


def member(payload: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, chunks=None) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    if chunks:
        body = b""
        o = 0
        for n in chunks:                      # Z_FULL_FLUSH forces several DEFLATE blocks in one member
            body += c.compress(payload[o:o + n]) + c.flush(zlib.Z_FULL_FLUSH)
            o += n
        body += c.compress(payload[o:]) + c.flush()
    else:
        body = c.compress(payload) + c.flush()
    total = 18 + len(body) + 8
    assert total <= 65536, total
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", total - 1) + body +
            struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


def corpus(rng):
    text = (b"@SIM:1:1000:9642:18772\nACGTTGCAACGT\n+\nIIIIHHHGGFFF\n" * 1200)[:60000]
    rnd = bytes(rng.getrandbits(8) for _ in range(30000))
    runs = b"".join(bytes([rng.randrange(4) + 65]) * rng.randrange(1, 400) for _ in range(300))[:65000]
    far = rnd[:20000] + bytes(12000) + rnd[:20000]          # matches at distance 32000
    cases = [b"", b"x", b"ab" * 3, text, rnd, runs, far, bytes(65280), bytes(range(256)) * 255]
    out = []
    for p in cases:
        out.append(member(p, 6))
        out.append(member(p, 1))
        out.append(member(p, 9))
        out.append(member(p[:40000], 0))                                        # stored blocks
        out.append(member(p, 6, zlib.Z_FIXED))                                  # fixed Huffman
        out.append(member(p, 6, zlib.Z_HUFFMAN_ONLY))
        out.append(member(p, 6, zlib.Z_RLE))
        if len(p) > 3000:
            out.append(member(p, 6, chunks=[1000, 1, 1500]))                    # several blocks / member
            out.append(member(p[:30000], 0, chunks=[7, 5000]))                  # stored + stored
            out.append(member(p, 6, zlib.Z_FIXED, chunks=[len(p) // 2]))
    return out


def test_inflate_block_types_vs_zlib(pkg, oracle):
    rng = random.Random(99)
    members = corpus(rng)
    rng.shuffle(members)
    data = b"".join(members) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    want, _ = oracle.bgzf_inflate_all(data)
    got, _ = pkg.bgzf_inflate(data)
    assert len(got) == len(want)
    assert got == want


def test_inflate_random_records(pkg, oracle):
    rng = random.Random(7)
    members = []
    for _ in range(300):
        n = rng.choice([1, 2, 17, 300, 5000, 30000, 65000])
        alphabet = rng.choice([2, 4, 20, 256])
        p = bytes(rng.randrange(alphabet) for _ in range(n))
        members.append(member(p, rng.choice([1, 4, 6, 9])))
    data = b"".join(members)
    want, _ = oracle.bgzf_inflate_all(data)
    got, _ = pkg.bgzf_inflate(data)
    assert got == want


def test_corrupt_members_are_rejected(pkg):
    good = member(b"hello world, hello world, hello world" * 50)
    bad_crc = bytearray(good)
    bad_crc[-8] ^= 0xFF
    with pytest.raises(pkg.BioscanError, match="CRC32"):
        pkg.bgzf_inflate(bytes(bad_crc))
    bad_body = bytearray(good)
    bad_body[25] ^= 0x55
    with pytest.raises(pkg.BioscanError):
        pkg.bgzf_inflate(bytes(bad_body))
    with pytest.raises(pkg.BioscanError, match="BGZF"):
        pkg.bgzf_inflate(b"\x1f\x8b\x08\x00" + bytes(40))


def test_member_whose_last_block_is_not_final_is_rejected(pkg):
    """A member whose only block has BFINAL = 0 and ends exactly where the payload ends: a decoder that goes on reads the
    trailer.  The payload is chosen so that its CRC32 field parses as one more block -- BFINAL = 1, fixed Huffman, END-OF-BLOCK
    at once (low ten bits 0b0000000_01_1) -- and ISIZE and CRC32 are right, so only "every block header lies inside the
    payload" rejects it (libdeflate: reading past the input is bad data; zlib: incomplete stream).  Found by
    tools/fuzz_k1_corrupt.py (seed 12, case 273: a flipped BFINAL bit)."""
    import deflate_build as db
    tail = bytes([200, 201, 202, 203, 204, 205])          # six 9-bit literals: 3 + 8 a + 9 * 6 + 7 bits = whole bytes
    k = 0
    while True:
        payload = b"not final %08d " % k + tail
        if zlib.crc32(payload) & 0x3FF == 0x003:
            break
        k += 1
    w = db.BitWriter()
    w.bits(0, 1)                                            # BFINAL = 0
    w.bits(1, 2)                                            # fixed Huffman
    for byte in payload:
        if byte < 144:
            w.code(0x30 + byte, 8)
        else:
            w.code(0x190 + byte - 144, 9)
    w.code(0, 7)                                            # END-OF-BLOCK
    assert w.n == 0                                         # the block ends on a byte boundary
    body = w.finish()
    with pytest.raises(zlib.error):
        zlib.decompress(body, -15)
    assert zlib.decompressobj(-15).decompress(body) == payload   # ... although every byte comes out
    total = 18 + len(body) + 8
    bad = (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", total - 1) + body +
           struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    with pytest.raises(pkg.BioscanError):
        pkg.bgzf_inflate(bad + eof)
    # the same block marked final is a valid member
    good_body = bytes([body[0] | 1]) + body[1:]
    good = bad[:18] + good_body + bad[18 + len(body):]
    got, _ = pkg.bgzf_inflate(good + eof)
    assert bytes(got) == payload


def test_inflate_handcrafted_huffman_codes(pkg, oracle):
    """Dynamic blocks whose code lengths zlib's encoder never picks (tests/deflate_build.py): complete but wildly
    skewed trees with 15-bit literal/length and distance codes (9-bit second-level distance tables), unused symbols
    with codes, one-symbol distance alphabets, several such blocks per member; every stream is checked with zlib first."""
    import deflate_build as db
    rng = random.Random(2024)
    members, max_code = [], 0
    for it in range(500):
        w, hist = db.BitWriter(), bytearray()
        nblk = rng.randint(1, 3)
        for b in range(nblk):
            toks, made = db.random_tokens(rng, rng.choice([1, 50, 3000, 20000]), alphabet=rng.choice([2, 16, 64, 256]), have=len(hist),
                                          match_prob=rng.choice([0.0, 0.2, 0.5, 0.9]))
            if len(hist) + made > 65000:
                toks = [('L', 1)]
            db.dynamic_block(w, toks, rng, b == nblk - 1, skew=rng.choice([0.3, 0.7, 0.95, 1.0]), extra_symbols=rng.choice([0, 5, 29]))
            db.apply_tokens(hist, toks)
        body, payload = w.finish(), bytes(hist)
        assert zlib.decompress(body, -15) == payload
        total = 18 + len(body) + 8
        if total > 65536:
            continue
        members.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", total - 1) + body +
                       struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))
    assert len(members) > 400
    data = b"".join(members) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    want, _ = oracle.bgzf_inflate_all(data)
    got, _ = pkg.bgzf_inflate(data)
    assert got == want


def test_invalid_code_length_sets_are_rejected(pkg):
    """Headers libdeflate (the reference's inflater) refuses even when the data never touches the bad part: incomplete
    and over-subscribed literal/length and distance codes.  The payload, ISIZE and CRC are consistent, so only the
    code-space check can catch them.  A single 1-bit distance code is the one incomplete code that must pass."""
    import deflate_build as db

    def member(lit, dist, toks):
        w = db.BitWriter()
        db.block_with_lengths(w, toks, lit, dist, True)
        hist = bytearray()
        db.apply_tokens(hist, toks)
        body, payload = w.finish(), bytes(hist)
        total = 18 + len(body) + 8
        return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", total - 1) + body +
                struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload))), body, payload

    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    lit = [0] * 258
    lit[65], lit[66], lit[256], lit[257] = 1, 2, 3, 3   # complete: 1/2 + 1/4 + 1/8 + 1/8
    toks = [('L', 65), ('L', 66), ('L', 65), ('M', 3, 1), ('L', 66)]
    good, body, payload = member(lit, [1], toks)    # one distance code of length 1: incomplete but legal
    assert zlib.decompress(body, -15) == payload
    assert pkg.bgzf_inflate(good + eof)[0] == payload
    cases = []
    bad = list(lit); bad[256] = 4                    # literal/length code incomplete: 1/2 + 1/4 + 1/16 + 1/8
    cases.append((bad, [1]))
    bad = list(lit); bad[67] = 2                     # over-subscribed: 1/2 + 1/4 + 1/4 + 1/8 + 1/8
    cases.append((bad, [1]))
    cases.append((lit, [2]))                         # a single distance code of length 2
    cases.append((lit, [1, 2]))                      # distance code incomplete: 1/2 + 1/4
    cases.append((lit, [1, 1, 1]))                   # distance code over-subscribed
    for l, d in cases:
        m, body, payload = member(l, d, toks)
        with pytest.raises(zlib.error):
            zlib.decompress(body, -15)
        with pytest.raises(pkg.BioscanError):
            pkg.bgzf_inflate(m + eof)


def test_header_prepass_opt_in():
    """BIOSCAN_K1_PREHEADERS=1 (K0, inflate_headers.hip: the first block header of every member parsed one member per lane
    ahead of K1; off by default): the cases of this file once more, in ONE child process started with
    the knob set (the library reads its knobs once per process)."""
    import subprocess
    import sys
    if os.environ.get("BIOSCAN_K1_PREHEADERS") == "1":
        pytest.skip("already running with the knob set")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BIOSCAN_K1_PREHEADERS="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_inflate_fuzz.py"), os.path.join(root, "tests", "test_gpu_bam_edge_cases.py"),
                          "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout
