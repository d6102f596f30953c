"""dev tool: differential fuzz of the BGZF deflate kernel (W2): random payloads of many shapes -> bioscan.bgzf_deflate -> every
member inflated by zlib (CRC32 + ISIZE checked) and by K1, compared with the input.  Skewed alphabets exercise the length
limit of the Huffman build (15 bits; 7 for the code-length code), tiny and empty members the fixed / stored fallbacks."""
import os, random, struct, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge._load_pkg(); pkg.load_library()


def members(data):
    out, o = [], 0
    while o < len(data):
        assert data[o:o + 4] == b"\x1f\x8b\x08\x04" and data[o + 12:o + 14] == b"BC", o
        bsize = struct.unpack_from("<H", data, o + 16)[0] + 1
        raw = zlib.decompressobj(-15)
        payload = raw.decompress(data[o + 18:o + bsize - 8]) + raw.flush()
        assert raw.eof and not raw.unused_data, o
        crc, isize = struct.unpack_from("<II", data, o + bsize - 8)
        assert isize == len(payload) and crc == zlib.crc32(payload), o
        out.append(payload)
        o += bsize
    return out


def gen(rng):
    kind = rng.randrange(9)
    n = rng.choice([0, 1, 2, 3, 100, 1000, 65279, 65280, 65281, 130560, rng.randrange(1, 400000)])
    if kind == 0:   # geometric alphabet: a few symbols carry almost everything (deep Huffman trees)
        p = rng.choice([0.5, 0.6, 0.7, 0.8])
        syms = list(range(256)); rng.shuffle(syms)
        out = bytearray()
        for _ in range(n):
            k = 0
            while rng.random() < p and k < 255: k += 1
            out.append(syms[k])
        return bytes(out)
    if kind == 1:   # Fibonacci frequencies
        fib = [1, 1]
        while len(fib) < rng.randrange(20, 34): fib.append(fib[-1] + fib[-2])
        pool = []
        for k, f in enumerate(fib): pool.extend([k] * min(f, 4000))
        return bytes(rng.choice(pool) for _ in range(n))
    if kind == 2:   # runs
        out = bytearray()
        while len(out) < n: out.extend(bytes([rng.randrange(256)]) * rng.randrange(1, rng.choice([4, 40, 400, 4000])))
        return bytes(out[:n])
    if kind == 3:   # repeated phrases at many distances
        words = [bytes(rng.randrange(256) for _ in range(rng.randrange(3, 40))) for _ in range(rng.randrange(2, 200))]
        out = bytearray()
        while len(out) < n: out.extend(rng.choice(words))
        return bytes(out[:n])
    if kind == 4:   # random bytes (stored blocks)
        return bytes(rng.getrandbits(8) for _ in range(n))
    if kind == 5:   # ACGT + qualities, BAM-like
        return bytes(rng.choice(b"ACGT") if i % 300 < 150 else 33 + min(40, max(2, int(rng.gauss(30, 6)))) for i in range(n))
    if kind == 6:   # every byte value equally often, no matches (256 nine-bit-ish codes)
        return bytes((i * 131 + (i >> 8) * 17) & 0xFF for i in range(n))
    if kind == 7:   # two symbols only, one rare
        return bytes(1 if rng.random() < 0.001 else 0 for _ in range(n))
    return bytes([rng.randrange(4)] * 1) * 0 + bytes(rng.choice(b"ab") for _ in range(n))


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    t0, cases, raw_bytes, comp_bytes = time.time(), 0, 0, 0
    types = {0: 0, 1: 0, 2: 0}
    while time.time() - t0 < seconds:
        data = gen(rng)
        comp, _ = pkg.bgzf_deflate(data, add_eof=True)
        mem = members(comp)
        assert b"".join(mem) == data, ("zlib", cases, len(data))
        assert all(len(m) <= 65280 for m in mem)
        back, _ = pkg.bgzf_inflate(comp)
        assert back == data, ("K1", cases, len(data))
        o = 0
        while o < len(comp):
            bs = struct.unpack_from("<H", comp, o + 16)[0] + 1
            if bs > 28: types[(comp[o + 18] >> 1) & 3] += 1
            o += bs
        cases += 1; raw_bytes += len(data); comp_bytes += len(comp)
        if cases % 50 == 0: print(f"{cases} cases, {raw_bytes / 1e6:.1f} MB, block types {types}", flush=True)
    print(f"OK: {cases} cases, {raw_bytes / 1e6:.1f} MB -> {comp_bytes / 1e6:.1f} MB, block types (stored, fixed, dynamic) {types}")


if __name__ == "__main__":
    main()
