#!/usr/bin/env python3
"""Sub-table space the dynamic Huffman codes of a BGZF file need (dev tool, CPU only): for every dynamic block, the
entries of the second-level tables behind a literal/length root of R_L bits and a distance root of R_D bits -- what a
reduced sub-table capacity in K1's LDS would have to hold.  usage: subtable_need.py FILE [n_members] [R_L] [R_D]"""
import struct, sys, zlib
from collections import Counter
sys.path.insert(0, __file__.rsplit('/', 1)[0])
from deflate_stats import Bits, build, dec, LEN_EXTRA, DIST_EXTRA

ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]

def sub_need(lens, root):
    """entries of the sub-tables: codes longer than root grouped by their root prefix, each group sized by its longest code"""
    cnt = Counter(l for l in lens if l)
    code, nxt = 0, {}
    for l in range(1, 16):
        code = (code + cnt.get(l - 1, 0)) << 1
        nxt[l] = code
    groups = {}
    for s, l in enumerate(lens):
        if l:
            c = nxt[l]; nxt[l] += 1
            if l > root:
                p = c >> (l - root)
                groups[p] = max(groups.get(p, 0), l - root)
    return sum(1 << w for w in groups.values())

def blocks(payload):
    b = Bits(payload + b"\0" * 8)
    while True:
        fin = b.take(1); typ = b.take(2)
        if typ == 0:
            b.pos = (b.pos + 7) & ~7
            n = b.take(16); b.take(16); b.pos += 8 * n
        elif typ == 1:
            yield None
            return  # (fixed blocks are not walked further here)
        else:
            hlit = b.take(5) + 257; hdist = b.take(5) + 1; hclen = b.take(4) + 4
            pl = [0] * 19
            for i in range(hclen): pl[ORDER[i]] = b.take(3)
            pt = build(pl)
            lens = []
            while len(lens) < hlit + hdist:
                s, _ = dec(b, pt)
                if s < 16: lens.append(s)
                elif s == 16: lens += [lens[-1]] * (3 + b.take(2))
                elif s == 17: lens += [0] * (3 + b.take(3))
                else: lens += [0] * (11 + b.take(7))
            ll, dl = lens[:hlit], lens[hlit:]
            yield ll, dl
            lt, dt = build(ll), build(dl)
            while True:
                s, _ = dec(b, lt)
                if s == 256: break
                if s > 256:
                    b.take(LEN_EXTRA[s - 257]); d, _ = dec(b, dt); b.take(DIST_EXTRA[d])
        if fin: return

def main():
    path = sys.argv[1]; nmem = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    rl = int(sys.argv[3]) if len(sys.argv) > 3 else 9; rd = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    data = open(path, 'rb').read(); o = 0; k = 0; needs = []
    while o < len(data) and k < nmem:
        xlen = data[o + 10] | data[o + 11] << 8
        bsize = (data[o + 16] | data[o + 17] << 8) + 1
        payload = data[o + 12 + xlen:o + bsize - 8]
        for blk in blocks(payload):
            if blk: needs.append((sub_need(blk[0], rl), sub_need(blk[1], rd), max(blk[0]), max(blk[1] or [0])))
        o += bsize; k += 1
    if not needs: print(path, "no dynamic blocks"); return
    print(path, "blocks", len(needs), "lit sub max/avg %d/%.0f" % (max(n[0] for n in needs), sum(n[0] for n in needs) / len(needs)),
          "dist sub max/avg %d/%.0f" % (max(n[1] for n in needs), sum(n[1] for n in needs) / len(needs)),
          "max code len lit %d dist %d" % (max(n[2] for n in needs), max(n[3] for n in needs)))

if __name__ == "__main__":
    main()
