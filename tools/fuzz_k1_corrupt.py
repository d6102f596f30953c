#!/usr/bin/env python3
"""Dev tool: K1 on damaged members -- random bit flips / truncations inside the DEFLATE body.  Every call must come back
(error or bytes), never hang or fault; when it returns bytes they must equal zlib's.  Run it on a V2_GUARD build first
(make EXTRA=-DV2_GUARD): the guards report out-of-range accesses instead of performing them.
usage: fuzz_k1_corrupt.py [n=300] [seed=7]"""
import os
import random
import struct
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conftest import load_pkg  # noqa: E402
import fuzz_k1 as fz  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    pkg = load_pkg()
    ok = err = same = 0
    for i in range(n):
        m = None
        while m is None or len(m) < 40:
            m = fz.zlib_member(rng) if rng.random() < 0.5 else fz.crafted_member(rng)
        body = bytearray(m[18:-8])
        for _ in range(rng.choice([1, 1, 2, 5, 40])):
            k = rng.randrange(len(body))
            body[k] ^= 1 << rng.randrange(8)
        if rng.random() < 0.15:
            body = body[:rng.randrange(1, len(body) + 1)]
        total = 18 + len(body) + 8
        bad = m[:16] + struct.pack("<H", total - 1) + bytes(body) + m[-8:]
        try:
            want = zlib.decompress(bytes(body), -15)
        except zlib.error:
            want = None
        try:
            got, _ = pkg.bgzf_inflate(bad + fz.EOF_BLOCK)
            ok += 1
            assert want is not None and got == want, ("GPU accepted a member zlib rejects or decodes differently", i)
            same += 1
        except pkg.BioscanError:
            err += 1
        if i % 50 == 49:
            print(i + 1, "done: rejected", err, "accepted", ok, flush=True)
    print("corrupt fuzz ok: rejected", err, "accepted and equal to zlib", same)


if __name__ == "__main__":
    main()
