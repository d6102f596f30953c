#!/usr/bin/env python3
"""Dev tool: extended K1 fuzz (more seeds than the test-suite affords): zlib-made members of every flavour plus
hand-built Huffman codes (tests/deflate_build.py), each batch inflated on the GPU and compared with zlib.
usage: fuzz_k1.py [n_batches=20] [first_seed=1000]"""
import os
import random
import struct
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg  # noqa: E402
import deflate_build as db  # noqa: E402

EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def wrap(body, payload):
    total = 18 + len(body) + 8
    if total > 65536:
        return None
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", total - 1) + body +
            struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


def zlib_member(rng):
    kind = rng.randrange(6)
    n = rng.choice([0, 1, 2, 100, 5000, 30000, 65000])
    if kind == 0:
        p = bytes(rng.getrandbits(8) for _ in range(min(n, 30000)))
    elif kind == 1:
        p = (b"@SIM:%d\nACGT\n+\nIIII\n" % rng.randrange(10 ** 6)) * (n // 20 + 1)
        p = p[:n]
    elif kind == 2:
        p = b"".join(bytes([rng.randrange(3) + 65]) * rng.randrange(1, 600) for _ in range(n // 100 + 1))[:n]
    elif kind == 3:
        a = bytes(rng.getrandbits(8) for _ in range(min(n // 3, 15000)))
        p = (a + bytes(rng.randrange(0, 20000)) + a)[:n]
    elif kind == 4:
        p = bytes(rng.choice(b"ACGTN") for _ in range(n))
    else:
        p = bytes(rng.randrange(30, 75) for _ in range(n))
    level = rng.choice([0, 1, 4, 6, 9])
    strat = rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
    c = zlib.compressobj(level, zlib.DEFLATED, -15, rng.choice([1, 8, 9]), strat)
    body = b""
    o = 0
    for _ in range(rng.randrange(0, 4)):
        k = rng.randrange(0, max(1, len(p) - o))
        body += c.compress(p[o:o + k]) + c.flush(rng.choice([zlib.Z_FULL_FLUSH, zlib.Z_SYNC_FLUSH]))
        o += k
    body += c.compress(p[o:]) + c.flush()
    return wrap(body, p)


def crafted_member(rng):
    w, hist = db.BitWriter(), bytearray()
    nblk = rng.randint(1, 4)
    for b in range(nblk):
        toks, made = db.random_tokens(rng, rng.choice([1, 10, 300, 3000, 20000]), alphabet=rng.choice([1, 2, 16, 64, 256]), have=len(hist),
                                      match_prob=rng.choice([0.0, 0.1, 0.5, 0.95]))
        if len(hist) + made > 65000:
            toks = [('L', 7)]
        db.dynamic_block(w, toks, rng, b == nblk - 1, max_len=rng.choice([7, 9, 10, 15]), skew=rng.choice([0.0, 0.5, 0.9, 1.0]),
                         extra_symbols=rng.choice([0, 3, 29]))
        db.apply_tokens(hist, toks)
    body, payload = w.finish(), bytes(hist)
    assert zlib.decompress(body, -15) == payload
    return wrap(body, payload)


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    pkg = load_pkg()
    total = 0
    for s in range(seed0, seed0 + n_batches):
        rng = random.Random(s)
        members = [m for m in (zlib_member(rng) if rng.random() < 0.5 else crafted_member(rng) for _ in range(300)) if m]
        data = b"".join(members) + EOF_BLOCK
        want = b"".join(zlib.decompress(m[18:-8], -15) for m in members)
        got, _ = pkg.bgzf_inflate(data)
        if got != want:
            print("MISMATCH seed", s)
            sys.exit(1)
        total += len(members)
        print("seed", s, "ok", len(members), "members", flush=True)
    print("fuzz ok:", total, "members")


if __name__ == "__main__":
    main()
