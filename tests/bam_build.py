"""Test helper: writes small BGZF-BAM files from Python records (SAM spec 4.2), so the tag-coercion corners of
sam_tag_io.rs:658-1036 can be exercised with values the reference's fixtures do not hold."""
import struct
import zlib

_SEQ_CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
_CIGAR_OP = {c: i for i, c in enumerate("MIDNSHP=X")}


def aux(tag: str, typ: str, value) -> bytes:
    """One aux field.  typ in A c C s S i I f Z H, or 'B<sub>' (value = list)."""
    t = tag.encode()
    if typ == "A":
        return t + b"A" + (value if isinstance(value, bytes) else bytes([value]))
    if typ in "cCsSiI":
        return t + typ.encode() + struct.pack("<" + {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I"}[typ], value)
    if typ == "f":
        if isinstance(value, int):  # raw bit pattern
            return t + b"f" + struct.pack("<I", value)
        return t + b"f" + struct.pack("<f", value)
    if typ in "ZH":
        return t + typ.encode() + (value if isinstance(value, bytes) else value.encode()) + b"\0"
    if typ[0] == "B":
        st = typ[1]
        fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[st]
        return t + b"B" + st.encode() + struct.pack("<i", len(value)) + struct.pack(f"<{len(value)}{fmt}", *value)
    raise ValueError(typ)


def record(name="r", refid=0, pos=100, mapq=30, flag=0, cigar=((10, "M"),), seq="ACGTACGTAC", qual=None, next_refid=-1,
           next_pos=-1, tlen=0, aux_bytes=b"") -> bytes:
    nm = name.encode() + b"\0"
    cg = b"".join(struct.pack("<I", (l << 4) | _CIGAR_OP[o]) for l, o in cigar)
    codes = [_SEQ_CODE[c] for c in seq] + [0]
    sq = bytes((codes[2 * i] << 4) | codes[2 * i + 1] for i in range((len(seq) + 1) // 2))
    ql = bytes(qual if qual is not None else [30] * len(seq))
    body = struct.pack("<iiBBHHHiiii", refid, pos, len(nm), mapq, 4680, len(cigar), flag, len(seq), next_refid, next_pos, tlen)
    body += nm + cg + sq + ql + aux_bytes
    return struct.pack("<i", len(body)) + body


def bgzf(payload: bytes, member: int = 60000) -> bytes:
    out = bytearray()
    chunks = [payload[i:i + member] for i in range(0, len(payload), member)] + [b""]
    for ch in chunks:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        d = c.compress(ch) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(d) + 25)
        out += d + struct.pack("<II", zlib.crc32(ch), len(ch))
    return bytes(out)


def bam(refs, records, text: str = None, member: int = 60000) -> bytes:
    """refs = [(name, length)]; records = iterable of record() bytes."""
    if text is None:
        text = "@HD\tVN:1.6\tSO:unsorted\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs)
    tb = text.encode()
    h = b"BAM\1" + struct.pack("<i", len(tb)) + tb + struct.pack("<i", len(refs))
    for n, l in refs:
        nb = n.encode() + b"\0"
        h += struct.pack("<i", len(nb)) + nb + struct.pack("<i", l)
    return bgzf(h + b"".join(records), member)
