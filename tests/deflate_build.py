"""Test helper: a DEFLATE *encoder* for dynamic-Huffman blocks with caller-chosen code lengths (RFC 1951 3.2.7), so
the inflate kernel meets codes zlib's own encoder never emits: 15-bit literal/length and distance codes (9-bit
second-level distance tables), complete but wildly skewed trees, one-symbol distance alphabets."""
import random

_LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
_DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
              8193, 12289, 16385, 24577]
_DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]
_CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


class BitWriter:
    def __init__(self):
        self.buf = bytearray()
        self.acc = 0
        self.n = 0

    def bits(self, v, n):           # LSB first
        self.acc |= (v & ((1 << n) - 1)) << self.n
        self.n += n
        while self.n >= 8:
            self.buf.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def code(self, c, n):           # Huffman codes are packed MSB first
        r = 0
        for _ in range(n):
            r = (r << 1) | (c & 1)
            c >>= 1
        self.bits(r, n)

    def finish(self):
        if self.n:
            self.buf.append(self.acc & 0xFF)
        return bytes(self.buf)


def canonical(lengths):
    """symbol -> (code, length) for the non-zero lengths (RFC 1951 3.2.2)."""
    count = [0] * 16
    for l in lengths:
        count[l] += 1
    count[0] = 0
    nxt, code = [0] * 16, 0
    for b in range(1, 16):
        code = (code + count[b - 1]) << 1
        nxt[b] = code
    out = {}
    for s, l in enumerate(lengths):
        if l:
            out[s] = (nxt[l], l)
            nxt[l] += 1
    return out


def random_complete_lengths(rng, symbols, max_len=15, skew=0.7):
    """Code lengths of a random full binary tree with one leaf per symbol (Kraft sum exactly 1); `skew` is the
    probability of splitting the deepest splittable leaf, which drives codes towards max_len."""
    n = len(symbols)
    if n == 1:
        return {symbols[0]: 1}
    max_len = min(15, max(max_len, (n - 1).bit_length()))
    while True:
        leaves = [0]
        while len(leaves) < n:
            cand = [i for i, d in enumerate(leaves) if d < max_len]
            if not cand:
                break                      # painted into a corner: every leaf is at max_len already
            i = max(cand, key=lambda k: leaves[k]) if rng.random() < skew else rng.choice(cand)
            d = leaves.pop(i)
            leaves += [d + 1, d + 1]
        if len(leaves) == n:
            break
        max_len = min(15, max_len + 1)
        skew = min(skew, 0.5)
    rng.shuffle(leaves)
    return dict(zip(symbols, leaves))


def len_symbol(length):
    for s in range(28, -1, -1):
        if length >= _LEN_BASE[s]:
            if s == 28 and length != 258:
                continue
            return s
    raise ValueError(length)


def dist_symbol(dist):
    for s in range(29, -1, -1):
        if dist >= _DIST_BASE[s]:
            return s
    raise ValueError(dist)


def dynamic_block(w: BitWriter, tokens, rng, final: bool, max_len=15, skew=0.7, extra_symbols=0):
    """tokens: ('L', byte) | ('M', length, distance).  Code lengths are random complete codes over the symbols used
    (+ `extra_symbols` unused ones, which lengthens the codes)."""
    lit_used, dist_used = {256}, set()
    for t in tokens:
        if t[0] == 'L':
            lit_used.add(t[1])
        else:
            lit_used.add(257 + len_symbol(t[1]))
            dist_used.add(dist_symbol(t[2]))
    pool = [s for s in range(286) if s not in lit_used]
    rng.shuffle(pool)
    lit_syms = sorted(lit_used | set(pool[:extra_symbols]))
    if len(lit_syms) == 1:
        lit_syms.append(pool[0] if pool[0] != 256 else pool[1])
        lit_syms.sort()
    dpool = [s for s in range(30) if s not in dist_used]
    rng.shuffle(dpool)
    dist_syms = sorted(dist_used | set(dpool[:min(extra_symbols, len(dpool))]))
    if not dist_syms:
        dist_syms = [0]
    ll = random_complete_lengths(rng, lit_syms, max_len, skew)
    dl = random_complete_lengths(rng, dist_syms, max_len, skew)  # a single distance code gets length 1 (allowed: incomplete)
    lit_lengths = [ll.get(s, 0) for s in range(max(lit_syms) + 1)]
    lit_lengths += [0] * (257 - len(lit_lengths))
    dist_lengths = [dl.get(s, 0) for s in range(max(dist_syms) + 1)]
    block_with_lengths(w, tokens, lit_lengths, dist_lengths, final)


def block_with_lengths(w: BitWriter, tokens, lit_lengths, dist_lengths, final: bool):
    """Dynamic block with the given code lengths, valid or not (lit_lengths: >= 257 entries incl. symbol 256)."""
    hlit, hdist = len(lit_lengths), len(dist_lengths)
    # code-length alphabet: symbols 0..15 literally (no repeat codes); a complete code: 13 x 4 bits + 6 x 5 bits
    cl_lengths = [0] * 19
    for k, s in enumerate(range(16)):
        cl_lengths[s] = 4 if k < 13 else 5
    cl_lengths[16] = cl_lengths[17] = cl_lengths[18] = 5
    cl_codes = canonical(cl_lengths)
    w.bits(1 if final else 0, 1)
    w.bits(2, 2)
    w.bits(hlit - 257, 5)
    w.bits(hdist - 1, 5)
    w.bits(19 - 4, 4)
    for s in _CL_ORDER:
        w.bits(cl_lengths[s], 3)
    for l in lit_lengths + dist_lengths:
        w.code(*cl_codes[l])
    lc, dc = canonical(lit_lengths), canonical(dist_lengths)
    for t in tokens:
        if t[0] == 'L':
            w.code(*lc[t[1]])
        else:
            s = len_symbol(t[1])
            w.code(*lc[257 + s])
            w.bits(t[1] - _LEN_BASE[s], _LEN_EXTRA[s])
            d = dist_symbol(t[2])
            w.code(*dc[d])
            w.bits(t[2] - _DIST_BASE[d], _DIST_EXTRA[d])
    w.code(*lc[256])


def random_tokens(rng, n_out, alphabet=64, match_prob=0.35, have=0):
    """A token list producing about n_out bytes after `have` bytes of history; returns (tokens, produced)."""
    toks, made = [], 0
    while made < n_out:
        pos = have + made
        if pos > 0 and rng.random() < match_prob:
            length = rng.choice([3, 4, 5, 8, 10, 11, 17, 31, 66, 130, 257, 258, rng.randint(3, 258)])
            dist = rng.choice([1, 2, 3, 4, 5, 16, 33, rng.randint(1, min(pos, 32768)), min(pos, 32768), min(pos, rng.choice([24577, 16385, 4097]))])
            dist = max(1, min(dist, pos, 32768))
            toks.append(('M', length, dist))
            made += length
        else:
            toks.append(('L', rng.randrange(alphabet)))
            made += 1
    return toks, made


def apply_tokens(history: bytearray, tokens):
    for t in tokens:
        if t[0] == 'L':
            history.append(t[1])
        else:
            for _ in range(t[1]):
                history.append(history[-t[2]])
