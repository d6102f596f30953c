#!/usr/bin/env python3
"""Symbol statistics of the DEFLATE streams of a BGZF file (dev tool, CPU only).

What K1's step count depends on: symbols per member, literal / match mix, code lengths, how many consecutive literal
pairs would fit a root table of R bits (the multi-literal table entry idea), blocks per member.
usage: deflate_stats.py FILE.bam [first_member [n_members]]
"""
import struct
import sys
from collections import Counter

LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
             8193, 12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


class Bits:
    def __init__(self, data):
        self.d = data
        self.pos = 0

    def take(self, n):
        v = 0
        for i in range(n):
            v |= ((self.d[(self.pos + i) >> 3] >> ((self.pos + i) & 7)) & 1) << i
        self.pos += n
        return v


def build(lens):
    """canonical Huffman: dict (len, code MSB-first) -> symbol"""
    cnt = Counter(l for l in lens if l)
    code, nxt = 0, {}
    for l in range(1, 16):
        code = (code + cnt.get(l - 1, 0)) << 1
        nxt[l] = code
    tab = {}
    for s, l in enumerate(lens):
        if l:
            tab[(l, nxt[l])] = s
            nxt[l] += 1
    return tab


def dec(b, tab):
    code = 0
    for l in range(1, 16):
        code = (code << 1) | b.take(1)
        s = tab.get((l, code))
        if s is not None:
            return s, l
    raise ValueError("bad code")


def member_stats(payload, st):
    b = Bits(payload + b"\0\0\0\0")
    nblk = 0
    while True:
        bfinal = b.take(1)
        bt = b.take(2)
        nblk += 1
        if bt == 0:
            b.pos = (b.pos + 7) & ~7
            ln = b.take(16)
            b.take(16)
            b.pos += 8 * ln
            st["stored_bytes"] += ln
        else:
            if bt == 1:
                ll = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
                dl = [5] * 30
            else:
                hlit, hdist, hclen = b.take(5) + 257, b.take(5) + 1, b.take(4) + 4
                order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
                pl = [0] * 19
                for i in range(hclen):
                    pl[order[i]] = b.take(3)
                pt = build(pl)
                lens = []
                while len(lens) < hlit + hdist:
                    s, _ = dec(b, pt)
                    if s < 16:
                        lens.append(s)
                    elif s == 16:
                        lens += [lens[-1]] * (3 + b.take(2))
                    elif s == 17:
                        lens += [0] * (3 + b.take(3))
                    else:
                        lens += [0] * (11 + b.take(7))
                ll, dl = lens[:hlit], lens[hlit:]
            lt, dt = build(ll), build(dl)
            body0 = b.pos
            prev_lit_len = None  # code length of the previous symbol if it was a literal
            run = 0
            while True:
                s, l = dec(b, lt)
                st["litlen_codelen"][l] += 1
                if s < 256:
                    st["lits"] += 1
                    st["lit_bits"] += l
                    if prev_lit_len is not None:
                        for R in (9, 10, 11, 12):
                            if prev_lit_len + l <= R:
                                st["pair_fit"][R] += 1
                        st["lit_after_lit"] += 1
                        prev_lit_len = None  # pairs are disjoint: a paired literal cannot start another pair
                    else:
                        prev_lit_len = l
                    run += 1
                elif s == 256:
                    break
                else:
                    if run:
                        st["lit_runs"][min(run, 64)] += 1
                    run = 0
                    prev_lit_len = None
                    k = s - 257
                    ml = LEN_BASE[k] + b.take(LEN_EXTRA[k])
                    ds, dlb = dec(b, dt)
                    dist = DIST_BASE[ds] + b.take(DIST_EXTRA[ds])
                    st["matches"] += 1
                    st["match_bytes"] += ml
                    st["match_bits"] += l + LEN_EXTRA[k] + dlb + DIST_EXTRA[ds]
                    st["dist_codelen"][dlb] += 1
                    st["mlen_hist"][min(ml, 64)] += 1
                    st["dist_hist"][dist.bit_length()] += 1
            st["block_bits"].append(b.pos - body0)
        if bfinal:
            break
    st["blocks"] += nblk
    st["members"] += 1


def main():
    path = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    data = open(path, "rb").read()
    st = Counter()
    for k in ("litlen_codelen", "dist_codelen", "pair_fit", "lit_runs", "mlen_hist", "dist_hist"):
        st[k] = Counter()
    st["block_bits"] = []
    o, idx = 0, 0
    while o + 28 <= len(data) and idx < first + n:
        bs = struct.unpack_from("<H", data, o + 16)[0] + 1
        xlen = struct.unpack_from("<H", data, o + 10)[0]
        if idx >= first and bs > 28:
            member_stats(data[o + 12 + xlen:o + bs - 8], st)
        o += bs
        idx += 1
    m = st["members"]
    sym = st["lits"] + st["matches"]
    print(f"members {m}, blocks/member {st['blocks'] / m:.2f}, block body bits avg {sum(st['block_bits']) / len(st['block_bits']):.0f}")
    print(f"symbols/member {sym / m:.0f} (+ distance lookups {st['matches'] / m:.0f}): literals {st['lits'] / m:.0f} ({st['lits'] / sym:.1%}), matches {st['matches'] / m:.0f}")
    print(f"output/member {(st['lits'] + st['match_bytes']) / m:.0f} B: literal bytes {st['lits'] / (st['lits'] + st['match_bytes']):.1%}, avg match {st['match_bytes'] / max(1, st['matches']):.1f} B")
    print(f"bits: literal avg {st['lit_bits'] / st['lits']:.2f}, match avg {st['match_bits'] / max(1, st['matches']):.2f}")
    print("litlen code length histogram:", dict(sorted(st["litlen_codelen"].items())))
    print("dist code length histogram:", dict(sorted(st["dist_codelen"].items())))
    lookups = st["lits"] + 2 * st["matches"]
    print(f"lookups (steps x lanes) now: {lookups / m:.0f} per member")
    for R in (9, 10, 11, 12):
        saved = st["pair_fit"][R]
        print(f"  root {R} bits: {saved / m:.0f} literal pairs fit -> lookups {(lookups - saved) / m:.0f} ({1 - saved / lookups:.1%} of now)")
    print("literal run lengths (between matches):", dict(sorted(st["lit_runs"].items())))
    print("match length hist (capped 64):", dict(sorted(st["mlen_hist"].items())))
    print("distance bit-length hist:", dict(sorted(st["dist_hist"].items())))


if __name__ == "__main__":
    main()
