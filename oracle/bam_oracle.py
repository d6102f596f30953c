"""CPU oracle for the BGZF -> BAM -> Arrow scan path.  TEST INFRASTRUCTURE ONLY.

This module is a plain-Python (zlib + struct + pyarrow) restatement of the
reference's partitioned BAM scan.  It is imported only by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg -- never by the
product path (the product path is the HIP library and fails loudly without it).

What it restates (paths relative to /root/reference/datafusion):
  * BGZF framing + inflate + CRC/ISIZE check ....... noodles-bgzf 0.49.0 (un-vendored
    dependency, Cargo.lock:3674-3864); call sites bio-format-bam/src/storage.rs:161-169,
    285-295.  Format per SAM spec 4.1 / RFC 1951-1952; inflate here is zlib.
  * BAM header + record field decode ................ noodles-bam 0.92.0 lazy Record accessors
    driven by bio-format-bam/src/physical_exec.rs:412-540 (sequential) and :939-1025 (indexed).
  * column builders / projection .................... bio-format-core/src/alignment_utils.rs:316-368,
    401-644, 695-701.
  * optional tags ................................... bio-format-core/src/sam_tag_io.rs:33-76,
    154-204, 658-1036.
  * BAI parsing, size estimates ..................... bio-format-bam/src/storage.rs:336-450.
  * balance_partitions .............................. bio-format-core/src/partition_balancer.rs:61-295.
  * region query + sub-region dedup + tails ......... bio-format-bam/src/physical_exec.rs:864-1372
    (noodles-csi 0.58.0 `BinningIndex::query`: reg2bins, linear-index min offset, chunk merge).
  * residual filters ................................ bio-format-core/src/record_filter.rs:57-283,
    bio-format-bam/src/storage.rs:456-494.
  * genomic filter extraction ....................... bio-format-core/src/genomic_filter.rs:51-350.
  * schema + metadata ............................... bio-format-bam/src/table_provider.rs:42-140,
    bio-format-core/src/metadata.rs:321-485.

PARITY PINNING: the reference (Rust + un-vendored noodles fork) cannot be built or run
in this environment, so this oracle is pinned by the reference's own test fixtures and
known-answer tests only: record counts per fixture / per chromosome / per partition
count (bio-format-bam/tests/indexed_read_test.rs:76,108,121,260-268,297-319;
indexed_read_large_test.rs), tag schema shapes (tests/tag_tests.rs), CIGAR KATs
(alignment_utils.rs:818-1017) and the balancer KATs (partition_balancer.rs:321-1005).
Per-value parity of `end`, `cigar`, `sequence`, `quality_scores`, `mate_*` read from a
committed BAM is NOT pinned by any reference test ("parity unpinned" for those values):
they follow the SAM/BAM specification and the rules quoted above.  One behaviour is an
explicit assumption (ZERO_SPAN_END below).
"""
from __future__ import annotations

import json
import struct
import zlib
from dataclasses import dataclass, field
from typing import Iterable, Optional

import pyarrow as pa

# ---------------------------------------------------------------------------------------
# Assumption (unverifiable offline): noodles-sam 0.87 `alignment_end` = start + span - 1
# with span 0 for an empty CIGAR, i.e. a placed read with zero reference span has
# end == start-1 (1-based) and `Position::new(0)` -> None when start == 1.  The
# reference's count tests (160/159/102 incl. zero-span placed-unmapped reads returned by
# the indexed query) prove alignment_end is Some(..) for those reads.
ZERO_SPAN_END = "start_minus_1"

CORE_FIELDS = [
    ("name", pa.utf8(), True),
    ("chrom", pa.utf8(), True),
    ("start", pa.uint32(), True),
    ("end", pa.uint32(), True),
    ("flags", pa.uint32(), False),
    ("cigar", pa.utf8(), False),
    ("mapping_quality", pa.uint32(), False),
    ("mate_chrom", pa.utf8(), True),
    ("mate_start", pa.uint32(), True),
    ("sequence", pa.utf8(), False),
    ("quality_scores", pa.utf8(), False),
    ("template_length", pa.int32(), False),
]
SEQ_LUT = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"
UNPLACED_SENTINEL = "*"  # bio-format-bam/src/storage.rs:23


# =======================================================================================
# BGZF
# =======================================================================================
def bgzf_blocks(data: bytes):
    """Yield (coffset, csize, payload_slice, crc, isize) for each BGZF member (SAM spec 4.1)."""
    o = 0
    n = len(data)
    while o < n:
        if n - o < 18:
            raise ValueError("truncated BGZF header")
        if data[o:o + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError(f"bad BGZF magic at {o}")
        xlen = struct.unpack_from("<H", data, o + 10)[0]
        # locate BC subfield
        p = o + 12
        bsize = None
        while p < o + 12 + xlen:
            si1, si2, slen = data[p], data[p + 1], struct.unpack_from("<H", data, p + 2)[0]
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", data, p + 4)[0] + 1
            p += 4 + slen
        if bsize is None:
            raise ValueError("BGZF block without BC subfield")
        crc, isize = struct.unpack_from("<II", data, o + bsize - 8)
        yield o, bsize, (o + 12 + xlen, o + bsize - 8), crc, isize
        o += bsize


def bgzf_inflate_all(data: bytes, check_crc: bool = True):
    """Returns (inflated bytes, block table [(coffset, uoffset, ulen)])."""
    out = []
    table = []
    uoff = 0
    for coff, _bs, (a, b), crc, isize in bgzf_blocks(data):
        raw = zlib.decompress(data[a:b], -15) if isize or b > a else b""
        if len(raw) != isize:
            raise ValueError("ISIZE mismatch")
        if check_crc and (zlib.crc32(raw) & 0xFFFFFFFF) != crc:
            raise ValueError("CRC mismatch")
        table.append((coff, uoff, isize))
        out.append(raw)
        uoff += isize
    return b"".join(out), table


# =======================================================================================
# BAM header
# =======================================================================================
@dataclass
class BamHeader:
    text: str
    ref_names: list
    ref_lengths: list
    first_record_offset: int


def parse_bam_header(u: bytes) -> BamHeader:
    if u[:4] != b"BAM\x01":
        raise ValueError("not a BAM file")
    l_text = struct.unpack_from("<i", u, 4)[0]
    text = u[8:8 + l_text].split(b"\x00")[0].decode("utf-8", "replace")
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", u, o)[0]
    o += 4
    names, lens = [], []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", u, o)[0]
        names.append(u[o + 4:o + 4 + l_name - 1].decode("utf-8", "replace"))
        lens.append(struct.unpack_from("<i", u, o + 4 + l_name)[0])
        o += 8 + l_name
    return BamHeader(text, names, lens, o)


def _sam_header_lines(text: str):
    for line in text.split("\n"):
        line = line.rstrip("\r")
        if len(line) >= 3 and line[0] == "@":
            kind = line[1:3]
            if kind == "CO":
                yield kind, line[4:] if len(line) > 3 else ""
            else:
                kv = []
                for f in line.split("\t")[1:]:
                    if len(f) >= 3 and f[2] == ":":
                        kv.append((f[:2], f[3:]))
                yield kind, kv


def extract_header_metadata(hdr: BamHeader) -> dict:
    """bio-format-core/src/metadata.rs:321-485 (JSON compared as parsed objects in tests:
    HashMap key order of `other_fields` is nondeterministic in the reference)."""
    md = {}
    sq, rg, pg, co = [], [], [], []
    for kind, val in _sam_header_lines(hdr.text):
        if kind == "HD":
            d = dict(val)
            if "VN" in d:
                md["bio.bam.file_format_version"] = d["VN"]
            if "SO" in d:
                md["bio.bam.sort_order"] = d["SO"]
            if "GO" in d:
                md["bio.bam.group_order"] = d["GO"]
            if "SS" in d:
                md["bio.bam.subsort_order"] = d["SS"]
        elif kind == "SQ":
            d = dict(val)
            e = {"name": d.get("SN", ""), "length": int(d.get("LN", "0"))}
            other = {k: v for k, v in val if k not in ("SN", "LN")}
            if other:
                e["other_fields"] = other
            sq.append(e)
        elif kind == "RG":
            d = dict(val)
            e = {"id": d.get("ID", "")}
            for k, nm in (("SM", "sample"), ("PL", "platform"), ("LB", "library"), ("DS", "description")):
                if k in d:
                    e[nm] = d[k]
            other = {k: v for k, v in val if k not in ("ID", "SM", "PL", "LB", "DS")}
            if other:
                e["other_fields"] = other
            rg.append(e)
        elif kind == "PG":
            d = dict(val)
            e = {"id": d.get("ID", "")}
            for k, nm in (("PN", "name"), ("VN", "version"), ("CL", "command_line")):
                if k in d:
                    e[nm] = d[k]
            other = {k: v for k, v in val if k not in ("ID", "PN", "VN", "CL")}
            if other:
                e["other_fields"] = other
            pg.append(e)
        elif kind == "CO":
            co.append(val)
    if not sq and hdr.ref_names:
        # noodles falls back to the binary reference list when the text has no @SQ
        sq = [{"name": n, "length": l} for n, l in zip(hdr.ref_names, hdr.ref_lengths)]
    if sq:
        md["bio.bam.reference_sequences"] = json.dumps(sq, separators=(",", ":"))
    if rg:
        md["bio.bam.read_groups"] = json.dumps(rg, separators=(",", ":"))
    if pg:
        md["bio.bam.program_info"] = json.dumps(pg, separators=(",", ":"))
    if co:
        md["bio.bam.comments"] = json.dumps(co, separators=(",", ":"))
    return md


# =======================================================================================
# Tag registry (SAM spec tags -> (sam type, arrow type, description)); tests/golden/sam_tag_registry.json
# is the data table (bio-format-core/src/tag_registry.rs:131-690).
# =======================================================================================
def _arrow_type_from_name(name: str) -> pa.DataType:
    m = {
        "Int32": pa.int32(), "UInt32": pa.uint32(), "Float32": pa.float32(), "Utf8": pa.utf8(),
        "List<Int8>": pa.list_(pa.field("item", pa.int8(), True)),
        "List<UInt8>": pa.list_(pa.field("item", pa.uint8(), True)),
        "List<Int16>": pa.list_(pa.field("item", pa.int16(), True)),
        "List<UInt16>": pa.list_(pa.field("item", pa.uint16(), True)),
        "List<Int32>": pa.list_(pa.field("item", pa.int32(), True)),
        "List<UInt32>": pa.list_(pa.field("item", pa.uint32(), True)),
        "List<Float32>": pa.list_(pa.field("item", pa.float32(), True)),
    }
    return m[name]


def _arrow_type_name(t: pa.DataType) -> str:
    if pa.types.is_list(t):
        inner = {pa.int8(): "Int8", pa.uint8(): "UInt8", pa.int16(): "Int16", pa.uint16(): "UInt16",
                 pa.int32(): "Int32", pa.uint32(): "UInt32", pa.float32(): "Float32"}[t.value_type]
        return f"List<{inner}>"
    return {pa.int32(): "Int32", pa.uint32(): "UInt32", pa.float32(): "Float32", pa.utf8(): "Utf8"}[t]


_REGISTRY = None


def known_tags() -> dict:
    global _REGISTRY
    if _REGISTRY is None:
        import os
        p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "sam_tag_registry.json")
        with open(p) as f:
            raw = json.load(f)
        _REGISTRY = {k: (v["sam_type"], _arrow_type_from_name(v["arrow_type"]), v["description"]) for k, v in raw.items()}
    return _REGISTRY


def sam_tag_type_to_arrow_type(c: str) -> pa.DataType:  # tag_registry.rs:757-768
    if c in "csi":
        return pa.int32()
    if c in "CSI":
        return pa.uint32()
    if c == "f":
        return pa.float32()
    if c == "B":
        return _arrow_type_from_name("List<Int32>")
    return pa.utf8()


_SUBTYPE = {"c": "Int8", "C": "UInt8", "s": "Int16", "S": "UInt16", "i": "Int32", "I": "UInt32", "f": "Float32"}


def parse_tag_type_hints(hints) -> dict:  # tag_registry.rs:698-755
    out = {}
    for h in hints or []:
        parts = h.split(":")
        if len(parts) == 2:
            tag, t = parts
            if len(t) != 1 or t == "B" or t not in "AcCsSiIfZH":
                raise ValueError(f"Invalid tag type hint '{h}'")
            out[tag] = (t, sam_tag_type_to_arrow_type(t))
        elif len(parts) == 3 and parts[1] == "B":
            tag, _, st = parts
            if len(st) != 1 or st not in _SUBTYPE:
                raise ValueError(f"Invalid tag type hint '{h}'")
            out[tag] = ("B", _arrow_type_from_name(f"List<{_SUBTYPE[st]}>"))
        else:
            raise ValueError(f"Invalid tag type hint '{h}'")
    return out


def format_sam_tag_type(sam_type: str, arrow_type: pa.DataType) -> str:
    """tag_registry.rs format_sam_tag_type: 'B' renders as 'B:<subtype>'."""
    if sam_type == "B" and pa.types.is_list(arrow_type):
        inv = {v: k for k, v in _SUBTYPE.items()}
        return "B:" + inv[_arrow_type_name(arrow_type)[5:-1]]
    return sam_type


# =======================================================================================
# Records
# =======================================================================================
@dataclass
class Rec:
    off: int          # offset of block_size field in the inflated stream
    size: int         # block_size
    refid: int
    pos: int
    l_read_name: int
    mapq: int
    n_cigar: int
    flag: int
    l_seq: int
    next_refid: int
    next_pos: int
    tlen: int


def iter_records(u: bytes, start: int, end: Optional[int] = None):
    o = start
    n = len(u) if end is None else end
    while o < n:
        if o + 4 > len(u):
            raise ValueError("truncated record length")
        bs = struct.unpack_from("<i", u, o)[0]
        if bs < 32 or o + 4 + bs > len(u):
            raise ValueError("truncated BAM record")
        refid, pos, lrn, mapq, _bin, ncig, flag, lseq, nref, npos, tlen = struct.unpack_from("<iiBBHHHiiii", u, o + 4)
        yield Rec(o, bs, refid, pos, lrn, mapq, ncig, flag, lseq, nref, npos, tlen)
        o += 4 + bs


def rec_name(u: bytes, r: Rec) -> str:
    raw = u[r.off + 36:r.off + 36 + r.l_read_name]
    if raw.endswith(b"\x00"):
        raw = raw[:-1]
    # noodles: name() is None for "*\0" -> reference appends "*" (physical_exec.rs:412-418)
    return raw.decode("utf-8", "replace")


def rec_cigar_ops(u: bytes, r: Rec):
    o = r.off + 36 + r.l_read_name
    return [struct.unpack_from("<I", u, o + 4 * i)[0] for i in range(r.n_cigar)]


def cigar_string(ops) -> str:  # alignment_utils.rs:695-701
    return "".join(f"{v >> 4}{CIGAR_OPS[v & 15]}" for v in ops)


def ref_span(ops) -> int:
    return sum(v >> 4 for v in ops if (v & 15) in (0, 2, 3, 7, 8))


def rec_end_1based(u: bytes, r: Rec) -> Optional[int]:
    """1-based inclusive alignment end (physical_exec.rs:448-463, 1316-1325)."""
    if r.pos < 0:
        return None
    start = r.pos + 1
    span = ref_span(rec_cigar_ops(u, r))
    end = start + span - 1
    if end < 1:
        return None
    return end


def rec_sequence(u: bytes, r: Rec) -> str:
    o = r.off + 36 + r.l_read_name + 4 * r.n_cigar
    raw = u[o:o + (r.l_seq + 1) // 2]
    out = []
    for i in range(r.l_seq):
        b = raw[i >> 1]
        out.append(SEQ_LUT[(b >> 4) if (i & 1) == 0 else (b & 15)])
    return "".join(out)


def rec_quality(u: bytes, r: Rec) -> str:
    o = r.off + 36 + r.l_read_name + 4 * r.n_cigar + (r.l_seq + 1) // 2
    # char::from(p + 33) on u8 (wrapping in release), pushed into a String (physical_exec.rs:482-487)
    return "".join(chr((q + 33) & 0xFF) for q in u[o:o + r.l_seq])


def rec_aux(u: bytes, r: Rec):
    """Yield (tag:str, type:str, value) in record order. Array values -> (subtype, list)."""
    o = r.off + 36 + r.l_read_name + 4 * r.n_cigar + (r.l_seq + 1) // 2 + r.l_seq
    end = r.off + 4 + r.size
    while o + 3 <= end:
        tag = u[o:o + 2].decode("latin-1")
        t = chr(u[o + 2])
        o += 3
        if t == "A":
            yield tag, t, u[o]; o += 1
        elif t == "c":
            yield tag, t, struct.unpack_from("<b", u, o)[0]; o += 1
        elif t == "C":
            yield tag, t, u[o]; o += 1
        elif t == "s":
            yield tag, t, struct.unpack_from("<h", u, o)[0]; o += 2
        elif t == "S":
            yield tag, t, struct.unpack_from("<H", u, o)[0]; o += 2
        elif t == "i":
            yield tag, t, struct.unpack_from("<i", u, o)[0]; o += 4
        elif t == "I":
            yield tag, t, struct.unpack_from("<I", u, o)[0]; o += 4
        elif t == "f":
            yield tag, t, struct.unpack_from("<f", u, o)[0]; o += 4
        elif t in "ZH":
            z = u.index(b"\x00", o)
            yield tag, t, u[o:z]; o = z + 1
        elif t == "B":
            st = chr(u[o]); cnt = struct.unpack_from("<i", u, o + 1)[0]; o += 5
            fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[st]
            vals = list(struct.unpack_from(f"<{cnt}{fmt}", u, o))
            o += cnt * struct.calcsize(fmt)
            yield tag, t, (st, vals)
        else:
            raise ValueError(f"bad aux type {t!r}")


# =======================================================================================
# Tag columns -> aux fields: the write side (build_tag_data sam_tag_io.rs:109-147, arrow_to_sam_tag_value :206-235,
# integer / float / string / hex / character / array conversions :237-656, parse_sam_tag_type tag_registry.rs:78-106) and
# the layout noodles-bam's encoder gives each value (SAM spec 4.2.4: tag[2] type[1] value; Z / H NUL-terminated; B = subtype,
# u32 count, elements).  Errors carry the reference's message.
# =======================================================================================
class TagWriteError(Exception):
    pass


_INT_RANGE = {"c": (-128, 127), "s": (-32768, 32767), "i": (-2 ** 31, 2 ** 31 - 1), "C": (0, 255), "S": (0, 65535), "I": (0, 2 ** 32 - 1)}
_INT_FMT = {"c": "<b", "s": "<h", "i": "<i", "C": "<B", "S": "<H", "I": "<I"}
_F32_MAX = 3.4028234663852886e38


def _rust_f64(v: float) -> str:
    if v != v:
        return "NaN"
    if v in (float("inf"), float("-inf")):
        return "inf" if v > 0 else "-inf"
    r = repr(float(v))
    if "e" in r or "E" in r:
        from decimal import Decimal
        r = format(Decimal(r), "f")
    if r.endswith(".0"):
        r = r[:-2]
    return r


def parse_sam_tag_type(spec: str):  # tag_registry.rs:78-106
    parts = spec.split(":")
    if len(parts) == 1:
        if len(parts[0]) != 1:
            raise TagWriteError(f"Invalid SAM tag type metadata: Invalid SAM tag type '{spec}': type must be a single character")
        return parts[0], None
    if len(parts) == 2 and parts[0] == "B":
        if len(parts[1]) != 1:
            raise TagWriteError(f"Invalid SAM tag type metadata: Invalid SAM tag array type '{spec}': subtype must be a single character")
        if parts[1] not in "cCsSiIf":
            raise TagWriteError(f"Invalid SAM tag type metadata: Unsupported SAM array subtype '{parts[1]}'")
        return "B", parts[1]
    raise TagWriteError(f"Invalid SAM tag type metadata: Invalid SAM tag type '{spec}': expected 'TYPE' or 'B:SUBTYPE'")


def _is_int_type(t: pa.DataType) -> bool:
    return pa.types.is_integer(t)


def _rust_type(t: pa.DataType) -> str:
    """the DataType's Debug form, for the types the tests use"""
    names = {pa.int8(): "Int8", pa.uint8(): "UInt8", pa.int16(): "Int16", pa.uint16(): "UInt16", pa.int32(): "Int32",
             pa.uint32(): "UInt32", pa.int64(): "Int64", pa.uint64(): "UInt64", pa.float32(): "Float32", pa.float64(): "Float64",
             pa.utf8(): "Utf8", pa.large_utf8(): "LargeUtf8", pa.binary(): "Binary", pa.bool_(): "Boolean"}
    if t in names:
        return names[t]
    if pa.types.is_list(t):
        return f"List({_rust_type(t.value_type)})"
    return str(t)


def tag_aux_bytes(tag: str, spec: str, arrow_type: pa.DataType, value) -> bytes:
    """One non-NULL value of a tag column as its aux field (b"" when the reference writes nothing)."""
    sam_type, sub = parse_sam_tag_type(spec)
    head = tag.encode("latin-1")
    if sam_type in _INT_RANGE:
        if not _is_int_type(arrow_type):
            raise TagWriteError(f"Tag value type mismatch for integer: {_rust_type(arrow_type)}")
        lo, hi = _INT_RANGE[sam_type]
        if not lo <= value <= hi:
            raise TagWriteError(f"Integer value {value} does not fit SAM type '{sam_type}'")
        return head + sam_type.encode() + struct.pack(_INT_FMT[sam_type], value)
    if sam_type == "f":
        if pa.types.is_float32(arrow_type):
            return head + b"f" + struct.pack("<f", value)
        if pa.types.is_float64(arrow_type):
            if value != value or abs(value) == float("inf") or not -_F32_MAX <= value <= _F32_MAX:
                raise TagWriteError(f"Float value {_rust_f64(value)} does not fit SAM type 'f'")
            return head + b"f" + struct.pack("<f", value)
        raise TagWriteError(f"Tag value type mismatch for float: {_rust_type(arrow_type)}")
    if sam_type == "Z":
        if not pa.types.is_string(arrow_type):
            raise TagWriteError(f"Tag value type mismatch for string: {_rust_type(arrow_type)}")
        return head + b"Z" + value.encode() + b"\0"
    if sam_type == "H":
        if not pa.types.is_string(arrow_type):
            raise TagWriteError(f"Tag value type mismatch for hex string: {_rust_type(arrow_type)}")
        norm = "".join(chr(ord(ch) - 32) if "a" <= ch <= "z" else ch for ch in value)
        if len(norm.encode()) % 2 or any(ch not in "0123456789ABCDEF" for ch in norm):
            raise TagWriteError(f"Invalid SAM hex tag value '{norm}'")
        return head + b"H" + norm.encode() + b"\0"
    if sam_type == "A":
        if pa.types.is_string(arrow_type):
            b = value.encode()
            if len(b) != 1 or b[0] >= 128:
                raise TagWriteError(f"Character tags must be a single ASCII byte, got '{value}'")
            return head + b"A" + b
        if _is_int_type(arrow_type):
            if not 0 <= value <= 255:
                raise TagWriteError(f"Character tag value {value} does not fit into a single byte")
            return head + b"A" + bytes([value])
        raise TagWriteError(f"Tag value type mismatch for character: {_rust_type(arrow_type)}")
    if sam_type == "B":
        if not pa.types.is_list(arrow_type):
            raise TagWriteError(f"Tag value type mismatch for array: {_rust_type(arrow_type)}")
        et = arrow_type.value_type
        if sub is None:  # sam_array_subtype_from_arrow_type (tag_registry.rs:48-59)
            sub = {pa.int8(): "c", pa.uint8(): "C", pa.int16(): "s", pa.uint16(): "S", pa.int32(): "i", pa.uint32(): "I",
                   pa.float32(): "f"}.get(et)
            if sub is None:
                raise TagWriteError(f"Unable to determine SAM array subtype for Arrow type {_rust_type(et)}")
        if any(v is None for v in value):
            raise TagWriteError("SAM array tags cannot contain null elements")
        out = head + b"B" + sub.encode() + struct.pack("<I", len(value))
        if sub == "f":
            if not pa.types.is_floating(et) or pa.types.is_float16(et):
                raise TagWriteError(f"Unsupported array element type for SAM subtype 'f': {_rust_type(et)}")
            for v in value:
                if pa.types.is_float64(et) and (v != v or abs(v) == float("inf") or not -_F32_MAX <= v <= _F32_MAX):
                    raise TagWriteError(f"Array element {_rust_f64(v)} does not fit SAM subtype 'f'")
                out += struct.pack("<f", v)
            return out
        if not _is_int_type(et):
            raise TagWriteError(f"Unsupported array element type for SAM subtype '{sub}': {_rust_type(et)}")
        lo, hi = _INT_RANGE[sub]
        for v in value:
            if not lo <= v <= hi:
                raise TagWriteError(f"Array element {v} does not fit SAM subtype '{sub}'")
            out += struct.pack(_INT_FMT[sub], v)
        return out
    if pa.types.is_string(arrow_type):  # any other type character on a string column (sam_tag_io.rs:227-233)
        return head + b"Z" + value.encode() + b"\0"
    return b""


def build_tag_data(batch: pa.RecordBatch, row: int) -> bytes:
    """Aux bytes of one row: the columns that carry bio.bam.tag.tag metadata, in schema order; NULL values are skipped."""
    out = b""
    for i, f in enumerate(batch.schema):
        md = f.metadata or {}
        if b"bio.bam.tag.tag" not in md:
            continue
        v = batch.column(i)[row]
        if not v.is_valid or len(f.name) != 2:
            continue
        spec = md.get(b"bio.bam.tag.type", b"Z").decode()
        out += tag_aux_bytes(f.name, spec, f.type, v.as_py())
    return out


# =======================================================================================
# Tag coercion (sam_tag_io.rs:658-1036)
# =======================================================================================
class TagError(Exception):
    pass


def _f32_to_string(v: float) -> str:
    """Rust `f32::to_string()` (core::fmt Display for f32 -> flt2dec::to_shortest_str): the shortest digit string whose
    value rounds back to the same f32 (interval ends included when the mantissa is even, as `decode` sets `inclusive`),
    the candidate closest to the exact value, printed positionally with no exponent; "NaN", "inf", "-inf", "-0".
    ASSUMPTION (core is not part of the reference tree): when the two shortest candidates are exactly equidistant
    (e.g. 343126.125f32 -> 343126.12 / 343126.13) the upper one is taken, which is what flt2dec's Grisu-with-Dragon-
    fallback does (`up && (!down || 2*mant >= scale)`); Ryu-style printers (numpy) round such ties to even instead.
    Restated as a search over digit counts with exact rationals (not the digit-generation loop the device code uses)."""
    import struct
    from fractions import Fraction
    bits = struct.unpack("<I", struct.pack("<f", v))[0]
    neg, be, frac = bits >> 31, (bits >> 23) & 0xFF, bits & 0x7FFFFF
    if be == 0xFF:
        return "NaN" if frac else ("-inf" if neg else "inf")
    sign = "-" if neg else ""
    if be == 0 and frac == 0:
        return sign + "0"
    m, e = ((frac | 0x800000), be - 150) if be else (frac, -149)
    two = Fraction(2) ** e
    val = m * two
    hi = val + two / 2
    lo = val - (two / 4 if (frac == 0 and be > 1) else two / 2)
    incl = (m & 1) == 0
    inside = (lambda x: lo <= x <= hi) if incl else (lambda x: lo < x < hi)
    k = 0  # 10^(k-1) <= val < 10^k
    while Fraction(10) ** k <= val:
        k += 1
    while Fraction(10) ** (k - 1) > val:
        k -= 1
    for n in range(1, 12):
        q = k - n
        unit = Fraction(10) ** q
        f = val / unit
        c0 = f.numerator // f.denominator
        c1 = c0 if f.denominator == 1 else c0 + 1
        ok0, ok1 = c0 > 0 and inside(c0 * unit), inside(c1 * unit)
        if not (ok0 or ok1):
            continue
        if ok0 and ok1:
            c = c1 if (c1 * unit - val) <= (val - c0 * unit) else c0
        else:
            c = c0 if ok0 else c1
        while c % 10 == 0:
            c //= 10
            q += 1
        ds = str(c)
        if q >= 0:
            return sign + ds + "0" * q
        if len(ds) <= -q:
            return sign + "0." + "0" * (-q - len(ds)) + ds
        return sign + ds[:q] + "." + ds[q:]
    raise AssertionError("f32 display: no round-tripping digit string found")


_LIST_RANGE = {"Int8": (-128, 127), "UInt8": (0, 255), "Int16": (-32768, 32767), "UInt16": (0, 65535),
               "Int32": (-2 ** 31, 2 ** 31 - 1), "UInt32": (0, 2 ** 32 - 1)}


def coerce_tag(t: str, value, arrow_type: pa.DataType):
    """Returns the python value appended to the column (None = NULL)."""
    tn = _arrow_type_name(arrow_type)
    if t in "csi" or t in "CSI":
        v = int(value)
        if tn == "Utf8":
            if 0 <= v <= 0x10FFFF and not (0xD800 <= v <= 0xDFFF):
                return chr(v)
            return str(v)
        if tn == "UInt32":
            if not (0 <= v <= 0xFFFFFFFF):
                raise TagError(f"integer value {v} does not fit UInt32")
            return v
        if tn == "Int32":
            if not (-2 ** 31 <= v <= 2 ** 31 - 1):
                raise TagError(f"integer value {v} does not fit Int32")
            return v
        # append_int on a non-int builder: OptionalField type mismatch
        raise TagError(f"tag value type mismatch: expected {tn}, got integer")
    if t == "f":
        if tn == "Utf8":
            return _f32_to_string(value)
        if tn == "Float32":
            return float(value)
        raise TagError(f"tag value type mismatch: expected {tn}, got float")
    if t in "ZH":
        try:
            s = bytes(value).decode("utf-8")
        except UnicodeDecodeError:
            return None
        if tn != "Utf8":
            raise TagError(f"tag value type mismatch: expected {tn}, got string")
        return s
    if t == "A":
        if tn == "UInt32" or tn == "Int32":
            return int(value)
        if tn != "Utf8":
            raise TagError(f"tag value type mismatch: expected {tn}, got character")
        return chr(value)
    if t == "B":
        st, vals = value
        if not tn.startswith("List<"):
            raise TagError(f"tag value type mismatch: expected {tn}, got array")
        inner = tn[5:-1]
        if inner == "Float32":
            if st != "f":
                raise TagError("tag value type mismatch: expected float list, got integer array")
            return [float(x) for x in vals]
        if st == "f":
            raise TagError(f"tag value type mismatch: expected {tn}, got float array")
        lo, hi = _LIST_RANGE[inner]
        for x in vals:
            if not (lo <= x <= hi):
                raise TagError(f"array element {x} does not fit {inner}")
        return list(vals)
    raise TagError("unknown tag type")


def infer_type_from_value(t: str, value):  # tag_registry.rs:772-792
    if t == "A":
        return "A", pa.utf8()
    if t in "cCsSi":
        return "i", pa.int32()
    if t == "I":
        return "I", pa.uint32()
    if t == "f":
        return "f", pa.float32()
    if t == "Z":
        return "Z", pa.utf8()
    if t == "H":
        return "H", pa.utf8()
    st, _ = value
    return "B", _arrow_type_from_name(f"List<{_SUBTYPE[st]}>")


# =======================================================================================
# BAI (SAM spec 5.2; noodles-bam bai reader keeps pseudo-bin 37450 as `metadata`, not in bins())
# =======================================================================================
@dataclass
class BaiRef:
    bins: dict            # bin id -> [(beg_voff, end_voff)]
    intervals: list       # linear index voffsets
    meta: Optional[tuple]  # (ref_beg, ref_end, n_mapped, n_unmapped)


@dataclass
class Bai:
    refs: list
    n_no_coor: Optional[int]


def parse_bai(data: bytes) -> Bai:
    if data[:4] != b"BAI\x01":
        raise ValueError("invalid BAI header")
    n_ref = struct.unpack_from("<i", data, 4)[0]
    o = 8
    refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", data, o)[0]
        o += 4
        bins = {}
        meta = None
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", data, o)
            o += 8
            chunks = [struct.unpack_from("<QQ", data, o + 16 * i) for i in range(n_chunk)]
            o += 16 * n_chunk
            if b == 37450:
                if n_chunk == 2:
                    meta = (chunks[0][0], chunks[0][1], chunks[1][0], chunks[1][1])
            else:
                bins[b] = chunks
        n_intv = struct.unpack_from("<i", data, o)[0]
        o += 4
        intervals = list(struct.unpack_from(f"<{n_intv}Q", data, o))
        o += 8 * n_intv
        refs.append(BaiRef(bins, intervals, meta))
    n_no_coor = struct.unpack_from("<Q", data, o)[0] if o + 8 <= len(data) else None
    return Bai(refs, n_no_coor)


def reg2bins(beg0: int, end0: int):
    """BAI binning scheme (min_shift 14, depth 5); beg0/end0 0-based half-open."""
    end0 -= 1
    bins = [0]
    for shift, base in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        bins.extend(range(base + (beg0 >> shift), base + (end0 >> shift) + 1))
    return bins


MAX_POS_BAI = 1 << 29


def bai_query_chunks(bai: Bai, ref_idx: int, start1: Optional[int], end1: Optional[int]):
    """noodles-csi 0.58 BinningIndex::query: chunks of overlapping bins, filtered by the
    linear-index min offset, sorted and merged."""
    s = start1 if start1 is not None else 1
    e = end1 if end1 is not None else MAX_POS_BAI
    if s > MAX_POS_BAI:
        raise ValueError("region start beyond max position")
    e = min(e, MAX_POS_BAI)
    ref = bai.refs[ref_idx]
    chunks = []
    for b in reg2bins(s - 1, e):
        chunks.extend(ref.bins.get(b, ()))
    i = (s - 1) >> 14
    min_off = ref.intervals[i] if i < len(ref.intervals) else 0
    chunks = sorted(c for c in chunks if c[1] > min_off)
    merged = []
    for c in chunks:
        if merged and c[0] <= merged[-1][1]:
            if c[1] > merged[-1][1]:
                merged[-1] = (merged[-1][0], c[1])
        else:
            merged.append(c)
    return merged


# =======================================================================================
# Planning: regions, estimates, balance_partitions
# =======================================================================================
@dataclass
class GenomicRegion:
    chrom: str
    start: Optional[int] = None
    end: Optional[int] = None
    unmapped_tail: bool = False


@dataclass
class RegionSizeEstimate:
    region: GenomicRegion
    estimated_bytes: int
    contig_length: Optional[int]
    unmapped_count: int = 0
    nonempty_bin_positions: list = field(default_factory=list)
    leaf_bin_span: int = 0


@dataclass
class PartitionAssignment:
    regions: list
    total_estimated_bytes: int


def estimate_sizes_from_bai(bai: Optional[Bai], regions, ref_names, ref_lengths):
    """bio-format-bam/src/storage.rs:336-436."""
    if bai is None:
        return [RegionSizeEstimate(r, 1, None, 0, [], 0) for r in regions]
    name_to_idx = {n: i for i, n in enumerate(ref_names)}
    out = []
    for r in regions:
        idx = name_to_idx.get(r.chrom)
        ref = bai.refs[idx] if idx is not None and idx < len(bai.refs) else None
        if ref is not None:
            mn, mx = (1 << 64) - 1, 0
            for chunks in ref.bins.values():
                for b, e in chunks:
                    mn = min(mn, b >> 16)
                    mx = max(mx, e >> 16)
            est = max(mx - mn, 0) if mx >= mn else 0
        else:
            est = 1
        clen = ref_lengths[idx] if idx is not None and idx < len(ref_lengths) else None
        if clen is not None and clen <= 0:
            clen = None
        unm = ref.meta[3] if ref is not None and ref.meta is not None else 0
        pos = sorted((b - 4681) * 16384 + 1 for b in ref.bins if 4681 <= b <= 37448) if ref is not None else []
        out.append(RegionSizeEstimate(r, est, clen, unm, pos, 16384))
    return out


def _partition_point(xs, pred):
    lo, hi = 0, len(xs)
    while lo < hi:
        mid = (lo + hi) // 2
        if pred(xs[mid]):
            lo = mid + 1
        else:
            hi = mid
    return lo


def balance_partitions(estimates, target_partitions: int):
    """bio-format-core/src/partition_balancer.rs:61-295 (integer arithmetic restated exactly)."""
    if not estimates:
        return []
    target = max(target_partitions, 1)
    if target == 1:
        return [PartitionAssignment([e.region for e in estimates], sum(e.estimated_bytes for e in estimates))]
    total = sum(e.estimated_bytes for e in estimates)
    if total == 0:
        nb = min(target, len(estimates))
        bins = [PartitionAssignment([], 0) for _ in range(nb)]
        for i, e in enumerate(estimates):
            bins[i % nb].regions.append(e.region)
        return bins
    eff_target = min(target, total)
    base = total // eff_target
    extra = total % eff_target

    def budget_for(i):
        return base + 1 if i < extra else base

    parts = [PartitionAssignment([], 0)]
    budget = budget_for(0)
    for est in estimates:
        remaining = est.estimated_bytes
        rs, re_, cl = est.region.start, est.region.end, est.contig_length
        if rs is not None and re_ is not None and re_ >= rs:
            eff_start, eff_end = rs, re_
        elif cl is not None and cl > 0:
            eff_start, eff_end = 1, cl
        else:
            eff_start, eff_end = 0, 0
        can_split = eff_end > 0 and eff_end >= eff_start
        pos = eff_start
        was_split = False
        if remaining == 0:
            mi = min(range(len(parts)), key=lambda i: (len(parts[i].regions), i))
            parts[mi].regions.append(est.region)
            continue
        while remaining > 0:
            if budget == 0 and len(parts) < eff_target:
                parts.append(PartitionAssignment([], 0))
                budget = budget_for(len(parts) - 1)
            is_last = len(parts) >= eff_target
            remaining_bp = eff_end - pos + 1 if (can_split and pos <= eff_end) else 0
            splittable = remaining_bp > 1
            if remaining <= budget or is_last or not splittable:
                if can_split and pos <= eff_end and was_split:
                    region = GenomicRegion(est.region.chrom, pos, None, False)
                else:
                    region = est.region
                p = parts[-1]
                p.regions.append(region)
                p.total_estimated_bytes += remaining
                budget = max(budget - remaining, 0)
                remaining = 0
            else:
                was_split = True
                nbp = est.nonempty_bin_positions
                if nbp and est.leaf_bin_span > 0:
                    r0 = _partition_point(nbp, lambda p_: p_ < pos)
                    r1 = _partition_point(nbp, lambda p_: p_ <= eff_end)
                    nbins = r1 - r0
                    if nbins > 1:
                        take = nbins * budget // remaining
                        take = max(1, min(take, nbins - 1))
                        bin_start = nbp[r0 + take - 1]
                        sub_end = min(bin_start + est.leaf_bin_span - 1, eff_end - 1)
                    else:
                        bp = remaining_bp * budget // remaining
                        sub_end = pos + max(1, min(bp, remaining_bp - 1)) - 1
                else:
                    bp = remaining_bp * budget // remaining
                    sub_end = pos + max(1, min(bp, remaining_bp - 1)) - 1
                p = parts[-1]
                p.regions.append(GenomicRegion(est.region.chrom, pos, sub_end, False))
                p.total_estimated_bytes += budget
                remaining -= budget
                pos = sub_end + 1
                if len(parts) < eff_target:
                    parts.append(PartitionAssignment([], 0))
                    budget = budget_for(len(parts) - 1)
                else:
                    budget = 0
        if est.unmapped_count > 0:
            p = parts[-1]
            p.regions.append(GenomicRegion(est.region.chrom, None, None, True))
            p.total_estimated_bytes += 1
            budget = max(budget - 1, 0)
    return [p for p in parts if p.regions]


# =======================================================================================
# Filters.  A filter is a tuple:
#   (col, op, value)            op in "=", "!=", "<", "<=", ">", ">="
#   (col, "between", (lo, hi))  /  (col, "not between", (lo, hi))
#   (col, "in", [v, ...])       /  (col, "not in", [v, ...])
# Top-level list = conjunction (DataFusion hands `scan` a list of conjuncts).
# =======================================================================================
def extract_genomic_regions(filters, zero_based: bool):
    """genomic_filter.rs:51-350 -> (regions, unsatisfiable)."""
    chroms, lo, hi = [], None, None
    for col, op, val in filters:
        if col == "chrom" and op == "=" and isinstance(val, str):
            chroms.append(val)
        elif col == "chrom" and op == "in":
            ex = [v for v in val if isinstance(v, str)]
            chroms.extend(ex)
        elif col == "start" and op in ("=", ">", ">=", "<", "<=") and isinstance(val, int) and val >= 0:
            v1 = val + 1 if zero_based else val
            if op == "=":
                lo = v1 if lo is None else max(lo, v1)
                hi = v1 if hi is None else min(hi, v1)
            elif op == ">":
                lo = v1 + 1 if lo is None else max(lo, v1 + 1)
            elif op == ">=":
                lo = v1 if lo is None else max(lo, v1)
            elif op == "<":
                u = max(v1 - 1, 0)
                hi = u if hi is None else min(hi, u)
            elif op == "<=":
                hi = v1 if hi is None else min(hi, v1)
        elif col == "end" and op in ("=", "<", "<=") and isinstance(val, int) and val >= 0:
            v1 = val
            if op == "<":
                v1 = max(val - 1, 0)
            hi = v1 if hi is None else min(hi, v1)
        elif col == "start" and op == "between" and all(isinstance(v, int) and v >= 0 for v in val):
            l1, h1 = (val[0] + 1, val[1] + 1) if zero_based else val
            lo = l1 if lo is None else max(lo, l1)
            hi = h1 if hi is None else min(hi, h1)
    chroms = sorted(set(chroms))
    unsat = lo is not None and hi is not None and lo > hi
    regions = [] if (not chroms or unsat) else [GenomicRegion(c, lo, hi, False) for c in chroms]
    return regions, unsat


_FIELD_TYPES = {"name": "s", "chrom": "s", "start": "n", "end": "n", "flags": "n", "cigar": "s",
                "mapping_quality": "n", "mate_chrom": "s", "mate_start": "n", "sequence": "s",
                "quality_scores": "s", "template_length": "n"}


def can_push_down_record_filter(f, schema_types=None) -> bool:
    """record_filter.rs:40-55, 285-356 for the BAM core schema (+ tag columns by arrow type)."""
    col, op, val = f
    types = dict(_FIELD_TYPES)
    if schema_types:
        types.update(schema_types)
    if col not in types:
        return False
    k = types[col]
    if op in ("=", "!="):
        return k in "sn"
    if op in ("<", "<=", ">", ">="):
        return k == "n"
    if op in ("between", "not between"):
        return k == "n"
    if op in ("in", "not in"):
        return k in "sn"
    return False


def _num(v):
    if isinstance(v, bool) or v is None:
        return None
    if isinstance(v, (int, float)):
        return float(v)
    return None


def evaluate_record_filters(fields: dict, filters, string_fields=("chrom",),
                            num_fields=("start", "end", "mapping_quality", "flags"), null_fields=()) -> bool:
    """record_filter.rs:57-283 with BamRecordFields (storage.rs:456-494): `chrom` is the only
    string field; start/end/mapping_quality/flags are u32 fields; anything else passes.
    null_fields: the accessor's `is_null_field` (record_filter.rs:87, 120, 153) -- a column the record reports as NULL
    fails every comparison, BETWEEN and [NOT] IN.  BamRecordFields never reports one (a missing value simply has no
    accessor and passes); VCF records do."""
    for col, op, val in filters:
        if col in null_fields and op in ("=", "!=", "<", "<=", ">", ">=", "between", "not between", "in", "not in"):
            return False
        if op in ("=", "!=", "<", "<=", ">", ">="):
            if val is None:
                return False
            sv = fields.get(col) if col in string_fields else None
            if sv is not None:
                if not isinstance(val, str):
                    continue
                if op == "=" and not (sv == val):
                    return False
                if op == "!=" and not (sv != val):
                    return False
                continue
            nv = fields.get(col) if col in num_fields else None
            if nv is None:
                continue
            lv = _num(val)
            if lv is None:
                continue
            rv = float(nv)
            ok = {"=": rv == lv, "!=": rv != lv, "<": rv < lv, "<=": rv <= lv, ">": rv > lv, ">=": rv >= lv}[op]
            if not ok:
                return False
        elif op in ("between", "not between"):
            lo, hi = val
            if lo is None or hi is None:
                return False
            nv = fields.get(col) if col in num_fields else None
            if nv is None:
                continue
            l, h = _num(lo), _num(hi)
            if l is None or h is None:
                continue
            b = l <= float(nv) <= h
            if (not b) if op == "between" else b:
                return False
        elif op in ("in", "not in"):
            neg = op == "not in"
            sv = fields.get(col) if col in string_fields else None
            nv = fields.get(col) if col in num_fields else None
            if sv is None and nv is None:
                continue
            saw_null, hit = False, False
            for lit in val:
                if sv is not None:
                    if lit is None:
                        saw_null = True
                    elif isinstance(lit, str) and lit == sv:
                        hit = True
                        break
                else:
                    l = _num(lit)
                    if l is None:
                        saw_null = True
                    elif float(nv) == l:
                        hit = True
                        break
            res = (not neg) if hit else ((not saw_null) and neg)
            if not res:
                return False
    return True


# =======================================================================================
# Table provider / exec mirror (names follow the reference)
def build_bam_header(metadata: dict, sort_on_write=None) -> str:
    """bio-format-bam/src/header_builder.rs:42-195 (build_bam_header) + the text noodles-sam's header writer produces for it,
    with the sort-order override of insert_into (table_provider.rs:1156-1164) when sort_on_write is True / False.
    `metadata`: the Arrow schema's metadata as str -> str.  Optional fields kept from other_fields: the standard tags
    header_builder.rs:192-250 maps; serde emits other_fields from a HashMap (unspecified order), here in key order."""
    md = dict(metadata)
    if sort_on_write is not None:
        md["bio.bam.sort_order"] = "coordinate" if sort_on_write else "unsorted"
    vn = "1.6"
    v = md.get("bio.bam.file_format_version")
    if v is not None:
        a, dot, b = v.partition(".")
        if dot and a.isascii() and b.isascii() and a.isdigit() and b.isdigit():
            vn = "%d.%d" % (int(a), int(b))
    out = "@HD\tVN:" + vn
    for key, tag in (("bio.bam.sort_order", "SO"), ("bio.bam.group_order", "GO"), ("bio.bam.subsort_order", "SS")):
        if key in md:
            out += "\t%s:%s" % (tag, md[key])
    out += "\n"

    def load(key, need):
        try:
            j = json.loads(md[key])
        except (KeyError, ValueError):
            return None
        if not isinstance(j, list) or not all(need(r) for r in j):
            return None
        return j

    def others(r, allowed):
        o = r.get("other_fields") if isinstance(r, dict) else None
        if not isinstance(o, dict):
            return ""
        return "".join("\t%s:%s" % (k, o[k]) for k in sorted(o) if k in allowed and isinstance(o[k], str))

    sq = load("bio.bam.reference_sequences", lambda r: isinstance(r, dict) and isinstance(r.get("name"), str)
              and isinstance(r.get("length"), int) and not isinstance(r.get("length"), bool) and r["length"] >= 0)
    for r in sq or []:
        if r["length"] == 0:
            raise ValueError("Reference sequence length cannot be zero")
        out += "@SQ\tSN:%s\tLN:%d%s\n" % (r["name"], r["length"], others(r, ("AH", "AN", "AS", "DS", "M5", "SP", "TP", "UR")))
    rg = load("bio.bam.read_groups", lambda r: isinstance(r, dict) and isinstance(r.get("id"), str))
    for r in rg or []:
        line = "@RG\tID:" + r["id"]
        for k, tag in (("sample", "SM"), ("platform", "PL"), ("library", "LB"), ("description", "DS")):
            if isinstance(r.get(k), str):
                line += "\t%s:%s" % (tag, r[k])
        out += line + others(r, ("BC", "CN", "DT", "FO", "KS", "PG", "PI", "PM", "PU")) + "\n"
    pg = load("bio.bam.program_info", lambda r: isinstance(r, dict) and isinstance(r.get("id"), str))
    for r in pg or []:
        line = "@PG\tID:" + r["id"]
        for k, tag in (("name", "PN"), ("version", "VN"), ("command_line", "CL")):
            if isinstance(r.get(k), str):
                line += "\t%s:%s" % (tag, r[k])
        out += line + others(r, ("PP", "DS")) + "\n"
    co = load("bio.bam.comments", lambda c: isinstance(c, str))
    for c in co or []:
        out += "@CO\t%s\n" % c
    return out


# =======================================================================================
class BamOracle:
    """Mirror of BamTableProvider::new + scan + BamExec::execute on the CPU."""

    def __init__(self, path: str, zero_based: bool = True, tag_fields=None, binary_cigar: bool = False,
                 infer_tag_types: bool = True, infer_tag_sample_size: int = 100, tag_type_hints=None,
                 index_path: Optional[str] = "auto"):
        import os
        self.path = path
        with open(path, "rb") as f:
            self.data = f.read()
        self.u, self.block_table = bgzf_inflate_all(self.data)
        self.hdr = parse_bam_header(self.u)
        self.zero_based = zero_based
        self.tag_fields = list(tag_fields) if tag_fields is not None else None
        self.binary_cigar = binary_cigar
        hints = parse_tag_type_hints(tag_type_hints) if tag_type_hints else None
        inferred = None
        if infer_tag_types and self.tag_fields:
            unknown = [t for t in self.tag_fields if t not in known_tags()]
            if unknown:
                inferred = self._discover(unknown, infer_tag_sample_size)
        self.schema = self._determine_schema(inferred, hints)
        # index discovery (index_utils.rs:44-77): <path>.bai, then <stem>.bai, then <path>.csi
        self.index_path = None
        if index_path == "auto":
            for cand in (path + ".bai", os.path.splitext(path)[0] + ".bai", path + ".csi"):
                if os.path.exists(cand):
                    self.index_path = cand
                    break
        else:
            self.index_path = index_path
        # The provider only keeps the path; every use reads it with `bam::bai::fs::read`.  A file that is not a BAI (the .csi
        # companion, a damaged index): unit size estimates in `scan` (storage.rs:344-360), no no-coor partition (:442-449),
        # and `IndexedBamReader::new` fails when an indexed partition is executed (storage.rs:286, physical_exec.rs:879-881).
        self.bai = None
        self.index_error = None
        if self.index_path:
            try:
                with open(self.index_path, "rb") as f:
                    self.bai = parse_bai(f.read())
            except (OSError, ValueError, struct.error) as e:
                self.bai, self.index_error = None, str(e)
        # block lookup for virtual offsets
        self._coff_to_uoff = {c: uo for c, uo, _ in self.block_table}
        self._coffs = [c for c, _, _ in self.block_table]

    # -- schema ---------------------------------------------------------------------
    def _discover(self, tags, sample_size):
        found = {}
        cnt = 0
        for r in iter_records(self.u, self.hdr.first_record_offset):
            if cnt >= sample_size:
                break
            first = {}
            for tag, t, v in rec_aux(self.u, r):
                first.setdefault(tag, (t, v))  # data.get(&tag) -> first occurrence
            for tg in tags:
                if tg in found or len(tg.encode()) != 2:
                    continue
                if tg in first:
                    found[tg] = infer_type_from_value(*first[tg])
            cnt += 1
        return found

    def _determine_schema(self, inferred, hints) -> pa.Schema:
        fields = []
        for name, typ, nullable in CORE_FIELDS:
            if name == "cigar" and self.binary_cigar:
                typ = pa.binary()
            fields.append(pa.field(name, typ, nullable))
        self.tag_types = []
        if self.tag_fields is not None:
            kt = known_tags()
            for tag in self.tag_fields:
                if inferred and tag in inferred:
                    st, at = inferred[tag]
                    desc = kt[tag][2] if tag in kt else f"Tag type discovered from file ({st})"
                elif hints and tag in hints:
                    st, at = hints[tag]
                    desc = kt[tag][2] if tag in kt else f"Tag type from user hint ({st})"
                elif tag in kt:
                    st, at, desc = kt[tag]
                else:
                    st, at, desc = "Z", pa.utf8(), "Unknown tag"
                md = {"bio.bam.tag.tag": tag, "bio.bam.tag.type": format_sam_tag_type(st, at),
                      "bio.bam.tag.description": desc}
                fields.append(pa.field(tag, at, True, metadata=md))
                self.tag_types.append(at)
        md = extract_header_metadata(self.hdr)
        md["bio.coordinate_system_zero_based"] = "true" if self.zero_based else "false"
        if self.binary_cigar:
            md["bio.bam.binary_cigar"] = "true"
        return pa.schema(fields, metadata=md)

    # -- planning -------------------------------------------------------------------
    def scan(self, projection=None, filters=(), target_partitions: int = 1):
        """Returns list of partition assignments (None = sequential single partition), residual filters."""
        filters = list(filters)
        if self.index_path:
            regions, unsat = extract_genomic_regions(filters, self.zero_based)
            if unsat:
                return [], []
            full = not regions
            if not regions and self.hdr.ref_names:
                regions = [GenomicRegion(n) for n in self.hdr.ref_names]
            if regions:
                est = estimate_sizes_from_bai(self.bai, regions, self.hdr.ref_names, self.hdr.ref_lengths)
                parts = balance_partitions(est, target_partitions)
                if full:
                    n = (self.bai.n_no_coor or 0) if self.bai is not None else 0
                    if n:
                        parts.append(PartitionAssignment([GenomicRegion(UNPLACED_SENTINEL, None, None, True)], max(n, 1)))
                tag_types = {}
                for f in self.schema:
                    if f.name not in _FIELD_TYPES:
                        t = f.type
                        tag_types[f.name] = "s" if t == pa.utf8() else ("n" if t in (pa.int32(), pa.uint32(), pa.float32()) else "x")
                residual = [f for f in filters if can_push_down_record_filter(f, tag_types)]
                return parts, residual
        return None, []

    # -- execution ------------------------------------------------------------------
    def _voff_to_uoff(self, voff: int) -> int:
        return self._coff_to_uoff[voff >> 16] + (voff & 0xFFFF)

    def _uoff_to_voff(self, uoff: int) -> int:
        import bisect
        uoffs = [uo for _, uo, _ in self.block_table]
        i = bisect.bisect_right(uoffs, uoff) - 1
        # a position at the exact end of block i is reported as the start of block i+1
        while i + 1 < len(self.block_table) and uoff >= self.block_table[i][1] + self.block_table[i][2]:
            i += 1
        c, uo, _ = self.block_table[i]
        return (c << 16) | (uoff - uo)

    def _start_out(self, pos):
        if pos < 0:
            return None
        return pos if self.zero_based else pos + 1

    def _row(self, r: Rec, flags, chrom_override="__rec__", start_override="__rec__", end_override="__rec__"):
        """accumulate_record_fields! (physical_exec.rs:939-1025) / sequential twin (:412-540)."""
        u = self.u
        row = {}
        names = self.hdr.ref_names
        if flags["name"]:
            row["name"] = rec_name(u, r)
        if flags["chrom"]:
            row["chrom"] = (names[r.refid] if r.refid >= 0 else None) if chrom_override == "__rec__" else chrom_override
        if flags["start"]:
            row["start"] = self._start_out(r.pos) if start_override == "__rec__" else start_override
        if flags["end"]:
            row["end"] = rec_end_1based(u, r) if end_override == "__rec__" else end_override
        if flags["flags"]:
            row["flags"] = r.flag
        if flags["cigar"]:
            ops = rec_cigar_ops(u, r)
            row["cigar"] = struct.pack(f"<{len(ops)}I", *ops) if self.binary_cigar else cigar_string(ops)
        if flags["mapping_quality"]:
            row["mapping_quality"] = r.mapq
        if flags["mate_chrom"]:
            row["mate_chrom"] = names[r.next_refid] if r.next_refid >= 0 else None
        if flags["mate_start"]:
            row["mate_start"] = self._start_out(r.next_pos)
        if flags["sequence"]:
            row["sequence"] = rec_sequence(u, r)
        if flags["quality_scores"]:
            row["quality_scores"] = rec_quality(u, r)
        if flags["template_length"]:
            row["template_length"] = r.tlen
        if flags["any_tag"] and self.tag_fields:
            vals = [[] for _ in self.tag_fields]
            idx = {t: i for i, t in enumerate(self.tag_fields)}
            for tag, t, v in rec_aux(u, r):
                if tag in idx:
                    vals[idx[tag]].append(coerce_tag(t, v, self.tag_types[idx[tag]]))
            for i, tg in enumerate(self.tag_fields):
                if len(vals[i]) > 1:
                    raise TagError("duplicate tag in record (reference appends twice: row misalignment)")
                row[tg] = vals[i][0] if vals[i] else None
        return row

    def _flags(self, projection):
        names = [f.name for f in self.schema]
        need = (lambda i: True) if projection is None else (lambda i: i in projection)
        fl = {n: need(i) for i, n in enumerate(names[:12])}
        fl["any_tag"] = True if projection is None else any(i >= 12 for i in projection)
        return fl

    def _rows_to_batches(self, rows, projection, batch_size):
        names = [f.name for f in self.schema]
        if projection is None:
            cols = list(range(len(names)))
        else:
            cols = list(projection)
        out_fields = [self.schema.field(i) for i in cols]
        out_schema = pa.schema(out_fields, metadata=self.schema.metadata)
        batches = []
        for s in range(0, len(rows), batch_size):
            chunk = rows[s:s + batch_size]
            if not cols:
                # zero-column batch with an explicit row count (alignment_utils.rs:360-363)
                rb = pa.RecordBatch.from_struct_array(pa.array([{}] * len(chunk), type=pa.struct([])))
                batches.append(rb.replace_schema_metadata(self.schema.metadata))
                continue
            arrays = []
            for i in cols:
                f = self.schema.field(i)
                arrays.append(pa.array([r.get(f.name) for r in chunk], type=f.type))
            batches.append(pa.RecordBatch.from_arrays(arrays, schema=out_schema))
        return out_schema, batches

    def execute_sequential(self, projection=None, batch_size: int = 8192):
        """get_local_bam_sync (physical_exec.rs:371-598)."""
        fl = self._flags(projection)
        rows = [self._row(r, fl) for r in iter_records(self.u, self.hdr.first_record_offset)]
        return self._rows_to_batches(rows, projection, batch_size)

    def _query(self, region: GenomicRegion):
        """IndexedBamReader::query + noodles filter (storage.rs:315-322)."""
        try:
            ref_idx = self.hdr.ref_names.index(region.chrom)
        except ValueError:
            raise ValueError(f"BAM region query failed: missing reference sequence {region.chrom}")
        s = region.start if region.start is not None else 1
        e = region.end if region.end is not None else None
        for beg, end in bai_query_chunks(self.bai, ref_idx, region.start, region.end):
            o = self._voff_to_uoff(beg)
            while o < len(self.u):
                if self._uoff_to_voff(o) >= end:
                    break
                bs = struct.unpack_from("<i", self.u, o)[0]
                refid, pos, lrn, mapq, _bin, ncig, flag, lseq, nref, npos, tlen = struct.unpack_from("<iiBBHHHiiii", self.u, o + 4)
                r = Rec(o, bs, refid, pos, lrn, mapq, ncig, flag, lseq, nref, npos, tlen)
                o += 4 + bs
                # intersects(): (Some(id), Some(start), Some(end)) required
                if r.refid != ref_idx or r.pos < 0:
                    continue
                end1 = rec_end_1based(self.u, r)
                if end1 is None:
                    continue
                start1 = r.pos + 1
                if not (s <= end1 and (e is None or start1 <= e)):
                    continue
                yield r

    def execute_partition(self, regions, projection=None, residual=(), batch_size: int = 8192):
        """get_indexed_stream (physical_exec.rs:864-1372)."""
        if self.index_error is not None:
            raise ValueError("Failed to open indexed BAM: " + self.index_error)
        fl = self._flags(projection)
        names = self.hdr.ref_names
        rows = []
        for region in regions:
            if region.unmapped_tail:
                if region.chrom == UNPLACED_SENTINEL:
                    if not (self.bai.n_no_coor or 0):
                        continue
                    for r in iter_records(self.u, self.hdr.first_record_offset):
                        if r.refid >= 0 or r.pos >= 0:
                            continue
                        if residual and not evaluate_record_filters(
                                {"chrom": None, "start": None, "end": None, "mapping_quality": r.mapq, "flags": r.flag}, residual):
                            continue
                        rows.append(self._row(r, fl, chrom_override=None, start_override=None, end_override=None))
                    continue
                ref_idx = names.index(region.chrom)
                ref = self.bai.refs[ref_idx]
                ends = [c[1] for ch in ref.bins.values() for c in ch]
                if ends:
                    seek = max(ends)
                else:
                    # last_first_record_start_position: max over refs of last linear-index entry
                    lasts = [rf.intervals[-1] for rf in self.bai.refs if rf.intervals]
                    seek = max(lasts) if lasts else 0
                o = self._voff_to_uoff(seek) if seek else self.hdr.first_record_offset
                seen = False
                for r in iter_records(self.u, o):
                    if r.refid == ref_idx:
                        seen = True
                    else:
                        if seen:
                            break
                        continue
                    if r.pos >= 0:
                        continue
                    if residual and not evaluate_record_filters(
                            {"chrom": region.chrom, "start": None, "end": None, "mapping_quality": r.mapq, "flags": r.flag}, residual):
                        continue
                    rows.append(self._row(r, fl, chrom_override=region.chrom, start_override=None, end_override=None))
                continue
            for r in self._query(region):
                pos1 = r.pos + 1
                if region.start is not None and pos1 < region.start:
                    continue
                if region.end is not None and pos1 > region.end:
                    continue
                start_val = self._start_out(r.pos)
                end_val = rec_end_1based(self.u, r)
                chrom = names[r.refid] if r.refid >= 0 else None
                if residual and not evaluate_record_filters(
                        {"chrom": chrom, "start": start_val, "end": end_val, "mapping_quality": r.mapq, "flags": r.flag}, residual):
                    continue
                rows.append(self._row(r, fl, chrom_override=chrom, start_override=start_val, end_override=end_val))
        return self._rows_to_batches(rows, projection, batch_size)
