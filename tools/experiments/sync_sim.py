#!/usr/bin/env python3
"""CPU simulation of K1's speculative sub-stream synchronisation (dev tool).

For every BGZF member: the true symbol chain of its (single dynamic) block, then for 64 equal sub-streams the decode a
speculative lane would do from `pre-roll` bits in front of its boundary (garbage END-OF-BLOCK / unassigned codes restart
at the literal/length root, as v3_sync does).  Reports how many lanes are NOT on the true chain when they cross their
boundary, and how far behind the boundary they re-join it (if they do inside their own sub-stream).
usage: sync_sim.py FILE.bam [first_member [n_members [preroll_bits ...]]]
"""
import struct
import sys
from collections import Counter

from deflate_stats import Bits, build, LEN_EXTRA, DIST_EXTRA


def fast_table(tab, maxbits=15):
    """(len, msb-first code) dict -> dict keyed by (lsb-first bits of up to 15) is too slow; build decode by length probing"""
    by_len = {}
    for (l, code), s in tab.items():
        by_len.setdefault(l, {})[code] = s
    return by_len


class Dec:
    def __init__(self, data, lt, dt):
        self.d = data
        self.lt = fast_table(lt)
        self.dt = fast_table(dt)

    def bit(self, p):
        return (self.d[p >> 3] >> (p & 7)) & 1

    def sym(self, p, t):
        code = 0
        for l in range(1, 16):
            code = (code << 1) | self.bit(p + l - 1)
            s = t.get(l, {}).get(code)
            if s is not None:
                return s, l
        return None, 0

    def step(self, p):
        """one litlen symbol (+ its distance) from bit p: returns (next p, kind) kind: 0 literal, 1 match, 2 eob, 3 bad"""
        s, l = self.sym(p, self.lt)
        if s is None or s > 285:
            return p + max(l, 1), 3
        p += l
        if s < 256:
            return p, 0
        if s == 256:
            return p, 2
        p += LEN_EXTRA[s - 257]
        ds, dl = self.sym(p, self.dt)
        if ds is None or ds > 29:
            return p + max(dl, 1), 3
        return p + dl + DIST_EXTRA[ds], 1


def member(payload, prerolls, st):
    b = Bits(payload + b"\0" * 8)
    bfinal, bt = b.take(1), b.take(2)
    if bt != 2 or not bfinal:
        st["skipped"] += 1
        return
    hlit, hdist, hclen = b.take(5) + 257, b.take(5) + 1, b.take(4) + 4
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    pl = [0] * 19
    for i in range(hclen):
        pl[order[i]] = b.take(3)
    pt = fast_table(build(pl))
    d0 = Dec(payload + b"\0" * 8, {}, {})
    lens = []
    while len(lens) < hlit + hdist:
        s, l = d0.sym(b.pos, pt)
        b.pos += l
        if s < 16:
            lens.append(s)
        elif s == 16:
            lens += [lens[-1]] * (3 + b.take(2))
        elif s == 17:
            lens += [0] * (3 + b.take(3))
        else:
            lens += [0] * (11 + b.take(7))
    dec = Dec(payload + b"\0" * 8, build(lens[:hlit]), build(lens[hlit:]))
    # true chain
    P0 = b.pos
    chain = set()
    p = P0
    while True:
        chain.add(p)
        p, k = dec.step(p)
        if k == 2:
            break
    end = p
    st["members"] += 1
    total = end - P0
    sub = ((total + 63) // 64 + 31) // 32 * 32
    for pre in prerolls:
        bad = 0
        for lane in range(1, 64):
            bnd = P0 + lane * sub
            if bnd >= end:
                break
            p = max(P0, bnd - pre)
            # decode to the first symbol start at / after bnd
            while p < bnd:
                p, k = dec.step(p)
            if p not in chain:
                bad += 1
                # how far until it re-joins
                q, n = p, 0
                while q < bnd + sub and q not in chain and q < end:
                    q, k = dec.step(q)
                    n += 1
                st[("rejoin", pre)][min(n, 400) // 25 * 25 if q in chain else -1] += 1
        st[("bad", pre)][bad] += 1
        st[("badsum", pre)] += bad


def main():
    path = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    prerolls = [int(x) for x in sys.argv[4:]] or [480]
    data = open(path, "rb").read()
    st = Counter()
    for pre in prerolls:
        st[("bad", pre)] = Counter()
        st[("rejoin", pre)] = Counter()
    o, idx = 0, 0
    while o + 28 <= len(data) and idx < first + n:
        bs = struct.unpack_from("<H", data, o + 16)[0] + 1
        xlen = struct.unpack_from("<H", data, o + 10)[0]
        if idx >= first and bs > 28:
            member(data[o + 12 + xlen:o + bs - 8], prerolls, st)
        o += bs
        idx += 1
    print("members", st["members"], "skipped", st["skipped"])
    for pre in prerolls:
        print(f"pre-roll {pre}: lanes off the chain at their boundary per member: avg {st[('badsum', pre)] / max(1, st['members']):.2f}; histogram {dict(sorted(st[('bad', pre)].items()))}")
        print(f"   steps until re-join (-1 = not inside its sub-stream): {dict(sorted(st[('rejoin', pre)].items()))}")


if __name__ == "__main__":
    main()
