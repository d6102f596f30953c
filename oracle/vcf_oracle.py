"""CPU oracle for the VCF scan path.  TEST INFRASTRUCTURE ONLY (see oracle/bam_oracle.py).

Restates (paths relative to /root/reference/datafusion):
  * schema determination ......................... bio-format-vcf/src/table_provider.rs:91-338
    (8 core fields, one column per INFO tag, single-sample FORMAT columns or one multi-sample
    `genotypes: Struct<tag: List<T>>`), info_to_arrow_type :1602-1626, format_to_arrow_type :370-396,
    single-sample column naming storage.rs:643-661, index contig names table_provider.rs:995-1076.
  * planning ....................................... table_provider.rs:1225-1462 (EmptyExec for limit 0 /
    unsatisfiable filters, filter regions or one region per index contig, estimate_sizes_from_tbi
    storage.rs:815-986, balance_partitions, residual filters) -- the balancer / filter code is shared
    with oracle/bam_oracle.py.
  * execution ...................................... physical_exec.rs:912-1198 (sequential) and :2747-3078
    (indexed): per-record core columns, get_variant_end :646-667, load_infos_single_pass :544-640,
    MultiSampleFormatBuilder::append_record :1634-1828 + append_list_of_samples :1860-2046,
    load_formats_single_pass :2265-2443, choose_effective_batch_size :81-137.
  * UDFs ........................................... udfs.rs:67-110 (list_avg), :606-650 (list_gte),
    list_lte (same loop with <=).

Record parsing is noodles-vcf 0.90.0 (un-vendored), restated from its published behaviour:
  - a line is split at tabs; ID / ALT / FILTER equal to "." are EMPTY collections, so the joined strings
    are "" (ASSUMPTION, noodles `Record::{ids,alternate_bases,filters}`); ALT alleles are re-joined with
    '|', IDs and filters with ';';
  - QUAL "." -> None, otherwise parsed as f32 and widened to f64;
  - INFO "." -> no fields; `key=value` is typed by the header definition of `key` (unknown keys: String,
    Number=1); a bare key is Flag when its type is Flag, otherwise the "missing value" error that the
    reference skips (physical_exec.rs:564, 642-644); value "." -> None; Number=1 -> scalar, other
    numbers -> array split at ',' with "." elements -> None; strings are percent-decoded;
  - `variant_end` = INFO END when present, else POS + len(REF) - 1 (ASSUMPTION: SVLEN / FORMAT LEN
    extensions of later VCF versions are not modelled; none of the fixtures or synthetic files use them);
  - samples: FORMAT keys zipped with the ':'-separated sample values (trailing values may be dropped),
    "." -> None, GT -> genotype re-rendered allele by allele (physical_exec.rs:1672-1697).
  - tabix query: bins overlapping the interval -> chunks, filtered by the linear-index minimum offset,
    sorted, merged; a record is yielded when its CHROM equals the region name and [POS, variant_end]
    intersects the interval (noodles-vcf io::reader::query).

PARITY PINNING: pinned by the values the reference's own tests assert --
tests/format_columns_test.rs:185-233, :257-330, :364-441 (multi-sample lists, sample subsets),
tests/info_missing_value_test.rs:36-154, tests/info_bare_key_test.rs:77-243, tests/special_char_info_test.rs,
tests/indexed_read_test.rs:99-232 (500/500/1000 rows, exactly one variant at 21:5000100),
indexed_read_large_test.rs:50-95, limit_and_indexed_projection_test.rs, udfs.rs unit tests :999-1162.
Adaptive re-tuning of the multi-sample batch size (physical_exec.rs:188-232) depends on Arrow builder
capacities and is not restated: multi-sample parity is on the concatenated rows of a partition.
"""
from __future__ import annotations

import json
import os
import struct
import zlib
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import pyarrow as pa

from bam_oracle import (GenomicRegion, PartitionAssignment, RegionSizeEstimate, balance_partitions,
                        bgzf_blocks, bgzf_inflate_all, extract_genomic_regions, evaluate_record_filters, reg2bins)

MAX_POS = 1 << 29


# =======================================================================================
# Header
# =======================================================================================
@dataclass
class FieldDefn:
    id: str
    number: str   # "0", "1", "2", ..., "A", "R", "G", "."
    type: str     # Integer / Float / Flag / Character / String
    description: str


@dataclass
class VcfHeader:
    file_format: str = "VCFv4.3"
    infos: dict = field(default_factory=dict)      # id -> FieldDefn (header order)
    formats: dict = field(default_factory=dict)
    filters: list = field(default_factory=list)    # [(id, description)]
    contigs: list = field(default_factory=list)    # [(id, length|None)]
    alts: list = field(default_factory=list)
    samples: list = field(default_factory=list)
    header_bytes: int = 0                           # bytes up to and including the #CHROM line


def _parse_struct_fields(body: str) -> dict:
    """`ID=x,Number=1,Description="a, b"` -> ordered dict; quoted values may contain commas and \\-escapes."""
    out, i, n = {}, 0, len(body)
    while i < n:
        j = body.find("=", i)
        if j < 0:
            break
        key = body[i:j].strip()
        i = j + 1
        if i < n and body[i] == '"':
            i += 1
            buf = []
            while i < n and body[i] != '"':
                if body[i] == "\\" and i + 1 < n:
                    i += 1
                buf.append(body[i])
                i += 1
            i += 1
            val = "".join(buf)
        else:
            k = body.find(",", i)
            if k < 0:
                k = n
            val = body[i:k]
            i = k
        out[key] = val
        if i < n and body[i] == ",":
            i += 1
    return out


def parse_vcf_header(text: str) -> VcfHeader:
    h = VcfHeader()
    pos = 0
    for line in text.split("\n"):
        ll = len(line) + 1
        line = line.rstrip("\r")
        if line.startswith("##"):
            pos += ll
            k, _, v = line[2:].partition("=")
            if k == "fileformat":
                h.file_format = v
            elif v.startswith("<") and v.endswith(">"):
                f = _parse_struct_fields(v[1:-1])
                if k == "INFO":
                    h.infos[f["ID"]] = FieldDefn(f["ID"], f.get("Number", "."), f.get("Type", "String"), f.get("Description", ""))
                elif k == "FORMAT":
                    h.formats[f["ID"]] = FieldDefn(f["ID"], f.get("Number", "."), f.get("Type", "String"), f.get("Description", ""))
                elif k == "FILTER":
                    h.filters.append((f["ID"], f.get("Description", "")))
                elif k == "contig":
                    ln = f.get("length")
                    h.contigs.append((f["ID"], int(ln) if ln is not None and ln.isdigit() else None))
                elif k == "ALT":
                    h.alts.append((f["ID"], f.get("Description", "")))
        elif line.startswith("#"):
            pos += ll
            cols = line.split("\t")
            h.samples = cols[9:] if len(cols) > 9 else []
            break
        else:
            break
    h.header_bytes = pos
    return h


def _scalar_arrow(ty: str) -> pa.DataType:
    return {"Integer": pa.int32(), "Float": pa.float32(), "Flag": pa.bool_()}.get(ty, pa.utf8())


def info_to_arrow_type(infos: dict, tag: str) -> pa.DataType:  # table_provider.rs:1602-1626
    d = infos.get(tag)
    if d is None:
        return pa.utf8()
    inner = _scalar_arrow(d.type)
    return inner if d.number in ("0", "1") else pa.list_(pa.field("item", inner, True))


def format_to_arrow_type(formats: dict, tag: str) -> pa.DataType:  # table_provider.rs:370-396
    if tag == "GT":
        return pa.utf8()
    d = formats.get(tag)
    if d is None:
        return pa.utf8()
    inner = pa.utf8() if d.type in ("String", "Character") else _scalar_arrow(d.type)
    return inner if d.number in ("0", "1") else pa.list_(pa.field("item", inner, True))


def resolve_single_sample_format_column_name(used: set, fid: str) -> str:  # storage.rs:643-661
    if fid not in used:
        return fid
    cand = f"fmt_{fid}"
    if cand in used:
        cand = f"format_{fid}"
    k = 2
    while cand in used:
        cand = f"format_{fid}_{k}"
        k += 1
    return cand


def _js(v) -> str:
    return json.dumps(v, separators=(",", ":"), ensure_ascii=False)


# =======================================================================================
# Tabix index
# =======================================================================================
@dataclass
class TbiRef:
    bins: dict
    intervals: list
    meta: Optional[tuple]


@dataclass
class Tbi:
    names: list
    refs: list
    n_no_coor: Optional[int]
    header: tuple  # (format, col_seq, col_beg, col_end, meta, skip)


def bgzf_decompress(data: bytes) -> bytes:
    return bgzf_inflate_all(data)[0]


def parse_tbi(raw: bytes) -> Tbi:
    d = bgzf_decompress(raw)
    if d[:4] != b"TBI\x01":
        raise ValueError("bad TBI magic")
    n_ref, fmt, col_seq, col_beg, col_end, meta, skip, l_nm = struct.unpack_from("<8i", d, 4)
    o = 36
    names = [s.decode() for s in d[o:o + l_nm].split(b"\x00")[:-1]] if l_nm else []
    o += l_nm
    refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", d, o)[0]
        o += 4
        bins, m = {}, None
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", d, o)
            o += 8
            chunks = [struct.unpack_from("<QQ", d, o + 16 * i) for i in range(n_chunk)]
            o += 16 * n_chunk
            if b == 37450:
                if n_chunk == 2:
                    m = (chunks[0][0], chunks[0][1], chunks[1][0], chunks[1][1])
            else:
                bins[b] = chunks
        n_intv = struct.unpack_from("<i", d, o)[0]
        o += 4
        intervals = list(struct.unpack_from(f"<{n_intv}Q", d, o))
        o += 8 * n_intv
        refs.append(TbiRef(bins, intervals, m))
    n_no_coor = struct.unpack_from("<Q", d, o)[0] if o + 8 <= len(d) else None
    return Tbi(names, refs, n_no_coor, (fmt, col_seq, col_beg, col_end, meta, skip))


def parse_csi_names(raw: bytes):
    """Reference names from a CSI header's tabix-style aux block (CSI spec: magic, min_shift, depth, l_aux, aux)."""
    d = bgzf_decompress(raw)
    if d[:4] != b"CSI\x01":
        raise ValueError("bad CSI magic")
    l_aux = struct.unpack_from("<i", d, 12)[0]
    if l_aux < 28:
        return []
    l_nm = struct.unpack_from("<i", d, 16 + 24)[0]
    return [s.decode() for s in d[44:44 + l_nm].split(b"\x00")[:-1]] if l_nm else []


def tbi_query_chunks(tbi: Tbi, ref_idx: int, start1: Optional[int], end1: Optional[int]):
    """noodles-csi BinningIndex::query for min_shift 14 / depth 5 (same scheme as BAI)."""
    s = start1 if start1 is not None else 1
    e = min(end1 if end1 is not None else MAX_POS, MAX_POS)
    if s > MAX_POS:
        raise ValueError("region start beyond max position")
    ref = tbi.refs[ref_idx]
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


_LEVELS = [(0, 1 << 29), (1, 1 << 26), (9, 1 << 23), (73, 1 << 20), (585, 1 << 17), (4681, 1 << 14)]


def estimate_sizes_from_tbi(tbi: Optional[Tbi], regions, contig_names, contig_lengths):
    """bio-format-vcf/src/storage.rs:815-986."""
    if tbi is None:
        return [RegionSizeEstimate(r, 1, None, 0, [], 0) for r in regions]
    index_name_to_idx = {n: i for i, n in enumerate(tbi.names)}
    contig_name_to_idx = {n: i for i, n in enumerate(contig_names)}
    length_by_name = {n: contig_lengths[i] for i, n in enumerate(contig_names)
                      if i < len(contig_lengths) and contig_lengths[i] > 0}
    out = []
    for r in regions:
        idx = index_name_to_idx.get(r.chrom)
        if idx is None:
            idx = contig_name_to_idx.get(r.chrom)
            if idx is not None and idx >= len(tbi.refs):
                idx = None
        ref = tbi.refs[idx] if idx is not None and idx < len(tbi.refs) else None
        if ref is not None:
            mn, mx = (1 << 64) - 1, 0
            for chunks in ref.bins.values():
                for b, e in chunks:
                    mn = min(mn, b >> 16)
                    mx = max(mx, e >> 16)
            est = max(mx - mn, 0)
        else:
            est = 1
        pos = sorted((b - 4681) * 16384 + 1 for b in ref.bins if 4681 <= b <= 37448) if ref is not None else []
        clen = length_by_name.get(r.chrom)
        if clen is None and pos:
            clen = pos[-1] + 16384 - 1
        if clen is None and ref is not None:
            best = None
            for b in ref.bins:
                for li in range(len(_LEVELS) - 1, -1, -1):
                    off, span = _LEVELS[li]
                    nxt = _LEVELS[li + 1][0] if li + 1 < len(_LEVELS) else 37449
                    if off <= b < nxt:
                        v = (b - off + 1) * span
                        best = v if best is None else max(best, v)
                        break
            clen = best
        out.append(RegionSizeEstimate(r, est, clen, 0, pos, 16384))
    return out


# =======================================================================================
# Record parsing
# =======================================================================================
class VcfError(Exception):
    pass


def _percent_decode(s: str) -> str:
    if "%" not in s:
        return s
    b = s.encode()
    out = bytearray()
    i = 0
    while i < len(b):
        if b[i] == 0x25 and i + 2 < len(b) and all(c in b"0123456789abcdefABCDEF" for c in b[i + 1:i + 3]):
            out.append(int(b[i + 1:i + 3], 16))
            i += 3
        else:
            out.append(b[i])
            i += 1
    return out.decode()


def parse_f32(s: str) -> float:
    """Rust `str::parse::<f32>` (correctly rounded, ties to even) -> python float holding the f32 value.  numpy's
    float32(str) goes through a double and rounds twice; this rounds once, with exact rational arithmetic."""
    from fractions import Fraction
    t = s.strip()
    low = t.lower().lstrip("+-")
    neg = t.startswith("-")
    if low in ("inf", "infinity"):
        return float("-inf") if neg else float("inf")
    if low == "nan":
        return float("nan")
    import re as _re
    if not _re.fullmatch(r"[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?", t):
        raise ValueError(f"invalid float literal: {s!r}")
    mant, _, ex = low.partition("e")
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0")
    if not digits:
        return -0.0 if neg else 0.0
    k = (int(ex) if ex else 0) - len(fp)
    if len(digits) + k > 60:
        return float("-inf") if neg else float("inf")
    if len(digits) + k < -60:
        return -0.0 if neg else 0.0
    v = Fraction(int(digits)) * (Fraction(10) ** k)
    # exponent e with 2^e <= v < 2^(e+1)
    e = v.numerator.bit_length() - v.denominator.bit_length()
    if Fraction(2) ** e > v:
        e -= 1
    if Fraction(2) ** (e + 1) <= v:
        e += 1
    q = max(e, -126) - 23                    # grid spacing 2^q (subnormals share 2^-149)
    scaled = v / (Fraction(2) ** q)
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1)):
        n += 1
    r = Fraction(n) * (Fraction(2) ** q)
    if r >= Fraction(2) ** 128:
        out = float("inf")
    else:
        out = float(r)
    return -out if neg else out


def parse_i32(s: str) -> int:
    v = int(s)
    if not (-(1 << 31) <= v < (1 << 31)) or not (s.lstrip("+-").isdigit()):
        raise VcfError(f"invalid integer: {s}")
    return v


_MISSING = object()   # bare non-flag key ("missing value" error that the reference skips)
FLAG = object()


class LazyList:
    """A list value noodles has typed but not walked: `Value::Array(..)` holds the text, and its elements are parsed only when the
    caller iterates them -- load_infos_single_pass does that for the keys it has a builder for (`values.iter().collect::<io::Result<_>>()`,
    physical_exec.rs:580-611), the FORMAT builder likewise (:1703-1760).  An element that does not parse is then the record's error;
    under a key without a builder nobody ever looks."""
    __slots__ = ("raw", "ty")

    def __init__(self, raw: str, ty: str):
        self.raw, self.ty = raw, ty

    def collect(self):
        parts = self.raw.split(",")
        if self.ty == "Integer":
            return [None if p == "." else parse_i32(p) for p in parts]
        if self.ty == "Float":
            return [None if p == "." else parse_f32(p) for p in parts]
        return [None if p == "." else _percent_decode(p) for p in parts]


class CharValue:
    """`Value::Character`: typed by noodles (exactly one character), but a value the reference has no builder arm for."""
    __slots__ = ("c",)

    def __init__(self, raw: str):
        if len(raw) != 1:
            raise VcfError(f"invalid character: {raw!r}")
        self.c = raw


def parse_info_fields(info: str, infos: dict, stop_at=None):
    """-> [(key, value)] in file order; value is FLAG, None ('.'), a scalar, a CharValue, a LazyList, or _MISSING.
    noodles' `info.iter(header)` types EVERY entry by the header as it goes (physical_exec.rs:561-571: an error in any entry
    is the record's error), whether or not the caller wants that key; a key the header does not declare is a String, Number=1.
    stop_at: stop behind the first entry with this key (`Info::get`, used for END)."""
    if info == "." or info == "":
        return []
    out = []
    for ent in info.split(";"):
        if ent == "":
            continue
        key, sep, raw = ent.partition("=")
        d = infos.get(key)
        number, ty = (d.number, d.type) if d is not None else ("1", "String")
        if not sep:
            out.append((key, FLAG if ty == "Flag" else _MISSING))
        elif raw == ".":
            out.append((key, None))
        elif ty == "Flag":
            raise VcfError(f"Error reading INFO field: invalid flag ({key})")
        elif number == "1":
            if ty == "Integer":
                out.append((key, parse_i32(raw)))
            elif ty == "Float":
                out.append((key, parse_f32(raw)))
            elif ty == "Character":
                out.append((key, CharValue(raw)))
            else:
                out.append((key, _percent_decode(raw)))
        else:
            out.append((key, LazyList(raw, ty)))
        if stop_at is not None and key == stop_at:
            break
    return out


class Rec:
    __slots__ = ("f", "chrom", "pos", "_info")

    def __init__(self, line: str):
        f = line.split("\t")
        if len(f) < 8:
            raise VcfError("VCF read error: invalid record")
        self.f = f
        self.chrom = f[0]
        if f[1] == "0":
            raise VcfError("Missing variant start")
        # noodles parses POS as usize (Rust's from_str: digits with an optional leading '+', no '-', no blanks, no '_')
        digits = f[1][1:] if f[1][:1] == "+" else f[1]
        if not digits or not digits.isascii() or not digits.isdigit():
            raise VcfError(f"VCF read error: invalid position {f[1]!r}")
        self.pos = int(digits)
        if self.pos > 0xFFFFFFFFFFFFFFFF:
            raise VcfError(f"VCF read error: invalid position {f[1]!r}")   # usize::from_str: number too large
        # (the reference casts the usize to u32 wherever a column or a filter takes it -- `get() as u32`,
        # physical_exec.rs:762, 663-665, 2875: a position that does not fit WRAPS; see `u32` below)
        self._info = None

    def info(self, infos):
        if self._info is None:
            self._info = parse_info_fields(self.f[7], infos)
        return self._info

    def variant_end(self, infos) -> int:
        # noodles `variant_end`: `info.get(header, "END")` walks the entries, typing each, up to the first END
        ents = self._info if self._info is not None else parse_info_fields(self.f[7], infos, stop_at="END")
        for k, v in ents:
            if k == "END":
                if isinstance(v, int):
                    return v
                break
        return self.pos + len(self.f[3]) - 1


def u32(v: int) -> int:
    """`as u32` of a position (usize): values beyond 2^32 - 1 wrap."""
    return v & 0xFFFFFFFF


def get_variant_end(rec: Rec, infos) -> int:  # physical_exec.rs:646-667 (`... .get() as u32`)
    ref, alt = rec.f[3], rec.f[4]
    alts = [] if alt == "." else alt.split(",")
    if len(ref) == 1 and len(alts) == 1 and ref in "ACGT" and alts[0] in ("A", "C", "G", "T"):
        return u32(rec.pos)
    return u32(rec.variant_end(infos))


def render_gt(raw: str) -> str:
    """noodles genotype iteration re-rendered (physical_exec.rs:1675-1694): alleles joined by their
    phasing; a leading phasing character (VCF >= 4.4) is dropped."""
    out, cur, first = [], "", True
    i = 0
    if raw and raw[0] in "/|":
        i = 1
    tok = ""
    sep_before = None
    for ch in raw[i:] + "\0":
        if ch in "/|\0":
            if tok == "" and ch == "\0" and not out:
                break
            if tok != "." and not tok.isdigit():
                raise VcfError(f"Error reading FORMAT field 'GT': invalid genotype {raw}")
            if not first:
                out.append(sep_before)
            out.append("." if tok == "." else str(int(tok)))
            first = False
            sep_before = ch
            tok = ""
        else:
            tok += ch
    return "".join(out)


def parse_sample_values(fmt_keys, sample: str, formats: dict, built=None):
    """-> [(key, value)]; value None for '.', python scalar / list otherwise; GT -> rendered string tagged.
    `sample.iter(header)` types every value of the sample (physical_exec.rs:1661-1666); a genotype and a list stay lazy and
    are only walked for the keys the FORMAT builder holds (`built`, None = all; :1664-1760) -- under another key they come
    back as None here, unlooked at."""
    if sample == "." or sample == "":
        vals = []
    else:
        vals = sample.split(":")
    out = []
    for key, raw in zip(fmt_keys, vals):
        if raw == ".":
            out.append((key, None))
            continue
        walked = built is None or key in built
        if key == "GT":
            out.append((key, ("GT", render_gt(raw))) if walked else (key, None))
            continue
        d = formats.get(key)
        number, ty = (d.number, d.type) if d is not None else ("1", "String")
        if number == "1":
            if ty == "Integer":
                out.append((key, parse_i32(raw)))
            elif ty == "Float":
                out.append((key, parse_f32(raw)))
            elif ty == "Character":
                out.append((key, CharValue(raw).c))   # (`SV::Character(c)` -> c.to_string(), :1699)
            else:
                out.append((key, _percent_decode(raw)))
        else:
            out.append((key, LazyList(raw, ty).collect() if walked else None))
    return out


def _wholly_missing(vals) -> bool:  # decoded_array_is_wholly_missing for text VCF (encoded == decoded length)
    return all(v is None for v in vals) and len(vals) <= 1


# =======================================================================================
# choose_effective_batch_size (physical_exec.rs:81-137)
# =======================================================================================
def choose_effective_batch_size(requested: int, any_format: bool, n_format_fields: int, n_selected: int, n_source: int) -> int:
    if not any_format or n_source <= 1 or n_selected == 0:
        return max(requested, 1)
    ffc = max(n_format_fields, 1)
    cells = n_selected * ffc
    if cells == 0:
        return max(requested, 1)
    bytes_per_sample = 16 + ffc * 8
    bytes_per_row = max(n_selected * bytes_per_sample, 1)
    by_cells = max(100_000 // cells, 1)
    by_bytes = max(8_000_000 // bytes_per_row, 1)
    eff = min(requested, by_cells, by_bytes)
    if by_cells >= 8 and by_bytes >= 8 and requested > 8:
        eff = max(eff, 8)
    return max(eff, 1)


# =======================================================================================
# Provider / exec mirror
# =======================================================================================
def _read_text_source(path: str):
    """-> (compression, decoded bytes, block table or None).  Block table rows: (coffset, csize, uoffset, ulen)."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) >= 18 and data[0] == 0x1F and data[1] == 0x8B and data[3] & 4 and data[12:14] == b"BC":
        blocks, out = [], []
        for coff, bsize, (a, b), crc, isize in bgzf_blocks(data):
            raw = zlib.decompress(data[a:b], -15) if isize or b > a else b""
            if len(raw) != isize or (zlib.crc32(raw) & 0xFFFFFFFF) != crc:
                raise ValueError("BGZF ISIZE / CRC mismatch")
            blocks.append((coff, bsize, sum(x[3] for x in blocks[-1:]) + (blocks[-1][2] if blocks else 0), isize))
            out.append(raw)
        return "bgzf", b"".join(out), blocks
    if data[:2] == b"\x1f\x8b":
        return "gzip", zlib.decompress(data, 47), None
    return "none", data, None


class VcfOracle:
    """Mirror of VcfTableProvider::new_with_samples + scan + VcfExec::execute on the CPU."""

    def __init__(self, path: str, info_fields=None, format_fields=None, samples=None, zero_based: bool = True,
                 index_path: Optional[str] = "auto"):
        self.path = path
        self.zero_based = zero_based
        self.compression, self.u, self.blocks = _read_text_source(path)
        # header: decode only the leading '#' lines
        end = 0
        while end < len(self.u) and self.u[end:end + 1] == b"#":
            nl = self.u.find(b"\n", end)
            end = len(self.u) if nl < 0 else nl + 1
        self.header = parse_vcf_header(self.u[:end].decode())
        self.data_start = end
        h = self.header
        self.info_fields = list(h.infos.keys()) if info_fields is None else list(info_fields)
        self.format_fields = list(h.formats.keys()) if format_fields is None else list(format_fields)
        self.source_samples = list(h.samples)
        if samples is None:
            self.samples = list(self.source_samples)
        else:  # MissingSamplePolicy::Ignore, first occurrence wins, request order kept
            seen, sel = set(), []
            for s in samples:
                if s in seen:
                    continue
                seen.add(s)
                if s in self.source_samples:
                    sel.append(s)
            self.samples = sel
        self.index_path = None
        if index_path == "auto":
            if self.compression == "bgzf":
                for cand in (path + ".tbi", path + ".csi"):
                    if os.path.exists(cand):
                        self.index_path = cand
                        break
        elif index_path:
            self.index_path = index_path
        # table_provider.rs:1011-1075: reference names from the TBI or CSI header; an index that cannot be read is only
        # logged there.  The scan then plans with unit estimates (storage.rs:826-840) and each partition fails when
        # IndexedVcfReader::new hands the file to the tabix reader (storage.rs:766; a CSI is rejected by it too).
        self.tbi = None
        self.index_error = None
        index_names = []
        if self.index_path:
            try:
                with open(self.index_path, "rb") as f:
                    raw = f.read()
                if self.index_path.lower().endswith(".csi"):
                    index_names = parse_csi_names(raw)
                    self.index_error = "invalid tabix header (CSI index)"
                else:
                    self.tbi = parse_tbi(raw)
                    index_names = list(self.tbi.names)
            except Exception as e:  # noqa: BLE001 - any read/parse failure is soft at open
                self.tbi = None
                self.index_error = str(e)
        self.contig_names = [c[0] for c in h.contigs]
        self.contig_lengths = [c[1] or 0 for c in h.contigs]
        self._indexed_names = None
        if index_names:
            by_name = {c[0]: c[1] for c in h.contigs if c[1] is not None}
            self.contig_lengths = [by_name.get(n, 0) for n in index_names]
            self.contig_names = list(index_names)
            self._indexed_names = list(index_names)
        self.schema = self._determine_schema()

    # ---- schema ------------------------------------------------------------------------------------
    def _format_field_meta(self, tag):
        d = self.header.formats.get(tag)
        md = {}
        if d is not None:
            md["bio.vcf.field.description"] = d.description
            md["bio.vcf.field.type"] = d.type
            md["bio.vcf.field.number"] = d.number
        md["bio.vcf.field.field_type"] = "FORMAT"
        md["bio.vcf.field.format_id"] = tag
        return md

    def _determine_schema(self) -> pa.Schema:
        h = self.header
        fields = [pa.field("chrom", pa.utf8(), False), pa.field("start", pa.uint32(), False),
                  pa.field("end", pa.uint32(), False), pa.field("id", pa.utf8(), True),
                  pa.field("ref", pa.utf8(), False), pa.field("alt", pa.utf8(), False),
                  pa.field("qual", pa.float64(), True), pa.field("filter", pa.utf8(), True)]
        for tag in self.info_fields:
            d = h.infos[tag]  # the reference unwraps: an unknown tag panics
            md = {"bio.vcf.field.description": d.description, "bio.vcf.field.type": d.type,
                  "bio.vcf.field.number": d.number, "bio.vcf.field.field_type": "INFO"}
            fields.append(pa.field(tag, info_to_arrow_type(h.infos, tag), d.type != "Flag", md))
        self.n_info = len(self.info_fields)
        self.format_columns = []  # single-sample column names
        if self.format_fields and self.samples:
            if len(self.source_samples) == 1:
                used = {f.name for f in fields}
                for tag in self.format_fields:
                    name = resolve_single_sample_format_column_name(used, tag)
                    used.add(name)
                    self.format_columns.append(name)
                    fields.append(pa.field(name, format_to_arrow_type(h.formats, tag), True, self._format_field_meta(tag)))
            else:
                kids = [pa.field(tag, pa.list_(pa.field("item", format_to_arrow_type(h.formats, tag), True)), True,
                                 self._format_field_meta(tag)) for tag in self.format_fields]
                md = {"bio.genotype.sample_names": _js(self.samples), "bio.vcf.genotypes.sample_names": _js(self.samples)}
                fields.append(pa.field("genotypes", pa.struct(kids), True, md))
        fmt_meta = {}
        for tag in sorted(set(self.format_fields)):
            d = h.formats.get(tag)
            if d is not None:
                fmt_meta[tag] = {"number": d.number, "type": d.type, "description": d.description}
        md = {
            "bio.coordinate_system_zero_based": "true" if self.zero_based else "false",
            "bio.vcf.file_format": h.file_format if h.file_format.startswith("VCFv") else h.file_format,
            "bio.vcf.filters": _js([{"id": i, "description": d} for i, d in h.filters]),
            "bio.vcf.contigs": _js([({"id": i, "length": l} if l is not None else {"id": i}) for i, l in h.contigs]),
            "bio.vcf.alternative_alleles": _js([{"id": i, "description": d} for i, d in h.alts]),
            "bio.vcf.samples": _js(self.samples),
            "bio.vcf.format_fields": _js(fmt_meta),
        }
        if self._indexed_names is not None:
            md["bio.vcf.contigs.indexed"] = _js(self._indexed_names)
        return pa.schema(fields, metadata=md)

    def supports_filters_pushdown(self, filters):
        out = []
        for f in filters:
            col = f[0]
            if self.index_path is not None and col in ("chrom", "start", "end") and (
                    f[1] in ("=", "!=", "<", "<=", ">", ">=") or (f[1] in ("between", "not between") and col != "chrom")
                    or (f[1] in ("in", "not in") and col == "chrom")):
                out.append("Inexact")
            elif self._can_push_down(f):
                out.append("Inexact")
            else:
                out.append("Unsupported")
        return out

    def _can_push_down(self, f) -> bool:  # record_filter.rs:40-55, 285-356 against this schema
        col, op, _ = f
        idx = self.schema.get_field_index(col)
        if idx < 0:
            return False
        t = self.schema.field(idx).type
        is_s = pa.types.is_string(t)
        is_n = t in (pa.uint32(), pa.int32(), pa.float32(), pa.float64())
        if op in ("=", "!="):
            return is_s or is_n
        if op in ("<", "<=", ">", ">=", "between", "not between"):
            return is_n
        if op in ("in", "not in"):
            return is_s or is_n
        return False

    # ---- planning ----------------------------------------------------------------------------------
    def scan(self, projection=None, filters=(), limit=None, target_partitions: int = 1):
        """-> dict(kind='empty'|'indexed'|'sequential', assignments, residual, projection, limit)."""
        plan = {"projection": None if projection is None else list(projection), "limit": limit,
                "residual": [], "assignments": None, "kind": "sequential"}
        if limit == 0:
            plan["kind"] = "empty"
            return plan
        if self.index_path is not None:
            regions, unsat = extract_genomic_regions(list(filters), self.zero_based)
            if unsat:
                plan["kind"] = "empty"
                return plan
            if not regions and self.contig_names:
                regions = [GenomicRegion(n) for n in self.contig_names]
            if regions:
                est = estimate_sizes_from_tbi(self.tbi, regions, self.contig_names, self.contig_lengths)
                plan["assignments"] = balance_partitions(est, target_partitions)
                plan["residual"] = [f for f in filters if self._can_push_down(f)]
                plan["kind"] = "indexed"
        return plan

    @staticmethod
    def num_partitions(plan) -> int:
        return 0 if plan["kind"] == "empty" else (len(plan["assignments"]) if plan["kind"] == "indexed" else 1)

    def projected_schema(self, projection):
        if projection is None:
            return self.schema
        return pa.schema([self.schema.field(i) for i in projection], metadata=self.schema.metadata)

    # ---- lines -------------------------------------------------------------------------------------
    def _lines_from(self, x: int, stop_abs: Optional[int] = None):
        """Yield (abs_offset, text) for lines starting at x while the line START is < stop_abs."""
        u, n = self.u, len(self.u)
        while x < n and (stop_abs is None or x < stop_abs):
            e = u.find(b"\n", x)
            nxt = n if e < 0 else e + 1
            e2 = n if e < 0 else e
            if e2 > x and u[e2 - 1] == 0x0D:
                e2 -= 1
            yield x, u[x:e2].decode()
            x = nxt

    def _voff_abs(self, voff: int) -> int:
        c, w = voff >> 16, voff & 0xFFFF
        for (co, cs, uo, ul) in self.blocks:
            if co == c:
                return uo + w
        if self.blocks and c == self.blocks[-1][0] + self.blocks[-1][1]:
            return len(self.u)
        raise VcfError(f"virtual offset {voff} does not address a block start")

    def _query(self, region: GenomicRegion):
        if region.chrom not in self.tbi.names:
            return  # "does not exist in reference sequences" -> region skipped (physical_exec.rs:2844-2850)
        idx = self.tbi.names.index(region.chrom)
        s = region.start if region.start is not None else 1
        e = region.end if region.end is not None else MAX_POS
        infos = self.header.infos
        for (cb, ce) in tbi_query_chunks(self.tbi, idx, region.start, region.end):
            for _, line in self._lines_from(self._voff_abs(cb), self._voff_abs(ce)):
                rec = self._rec(line)
                if rec.chrom != region.chrom:
                    continue
                if rec.pos <= e and rec.variant_end(infos) >= s:
                    yield rec

    # ---- rows --------------------------------------------------------------------------------------
    def _flags(self, projection):
        has = (lambda i: True) if projection is None else (lambda i: i in projection)
        fl = {k: has(i) for i, k in enumerate(("chrom", "start", "end", "id", "ref", "alt", "qual", "filter"))}
        fl["any_info"] = projection is None or any(8 <= i < 8 + self.n_info for i in projection)
        fl["any_format"] = projection is None or any(i >= 8 + self.n_info for i in projection)
        return fl

    def _rec(self, line: str) -> Rec:
        return Rec(line)

    def _core_row(self, rec: Rec, fl=None):
        """The eight core columns, each only when the projection holds it (physical_exec.rs:800-822: noodles' record is lazy,
        so a QUAL that does not parse, or an INFO entry in front of END, is an error only for a scan that asks for it)."""
        f = rec.f
        need = (lambda k: True) if fl is None else (lambda k: fl[k])
        row = {"chrom": rec.chrom, "start": u32(u32(rec.pos) - 1) if self.zero_based else u32(rec.pos),   # (`start_pos_1based - 1` on a u32)
               "id": "" if f[2] == "." else f[2], "ref": f[3], "alt": "" if f[4] == "." else f[4].replace(",", "|"),
               "filter": "" if f[6] == "." else f[6]}
        row["end"] = get_variant_end(rec, self.header.infos) if need("end") else None
        row["qual"] = (None if f[5] == "." else parse_f32(f[5])) if need("qual") else None
        return row

    def _info_row(self, rec: Rec, projected=None):
        """load_infos_single_pass -> list of per-field python values in self.info_fields order.
        projected: indices (into info_fields) of the INFO columns the batch will hold (None = all).  A key that occurs twice
        is appended twice to its builder (:572-575), which leaves THAT column one row longer than the others: the batch
        cannot be assembled if the column is part of it (RecordBatch::try_new), and nobody notices if it is not."""
        h = self.header
        types = [info_to_arrow_type(h.infos, t) for t in self.info_fields]
        idx_of = {t: i for i, t in enumerate(self.info_fields)}
        vals = [None] * len(types)
        populated = [False] * len(types)
        for key, v in rec.info(h.infos):
            if v is _MISSING:
                continue
            i = idx_of.get(key)
            if i is None:
                continue       # no builder: a list under this key is never walked
            if isinstance(v, CharValue) or (isinstance(v, LazyList) and v.ty == "Character"):
                raise VcfError(f"Unsupported INFO value type for field '{key}'")   # (:632-636)
            if isinstance(v, LazyList):
                v = v.collect()
            if populated[i]:
                if projected is None or i in projected:
                    raise VcfError(f"duplicate INFO key {key} (the reference appends twice and misaligns rows)")
                continue
            populated[i] = True
            t = types[i]
            if v is FLAG:
                if t != pa.bool_():
                    raise VcfError("Expected BooleanBuilder")
                vals[i] = True
            elif v is None:
                vals[i] = False if t == pa.bool_() else None
            elif isinstance(v, list):
                if not pa.types.is_list(t):
                    raise VcfError(f"INFO array value for scalar column {key}")
                vals[i] = v
            else:
                if pa.types.is_list(t):  # append_int / append_float on an array builder: one-element list
                    if isinstance(v, str):
                        raise VcfError("Expected Utf8Builder")
                    vals[i] = [v]
                else:
                    vals[i] = v
        for i, t in enumerate(types):
            if not populated[i] and t == pa.bool_():
                vals[i] = False
        return vals

    def _sel_indices(self):
        src = {n: i for i, n in enumerate(self.source_samples)}
        return [src[n] for n in self.samples if n in src]

    def _format_multi_row(self, rec: Rec):
        """MultiSampleFormatBuilder::append_record -> per FORMAT field a list of per-sample cells."""
        h = self.header
        nf = len(self.format_fields)
        f = rec.f
        keys = f[8].split(":") if len(f) > 8 and f[8] not in ("", ".") else []
        samples = f[9:]
        out_of_header = {hi: oi for oi, hi in enumerate(self._sel_indices())}
        ns = len(self.samples)
        cells = [[None] * nf for _ in range(ns)]
        field_idx = {t: i for i, t in enumerate(self.format_fields)}
        remaining = ns
        for hi, s in enumerate(samples):
            oi = out_of_header.get(hi)
            if oi is None:
                continue
            for key, v in parse_sample_values(keys, s, h.formats, set(self.format_fields)):
                i = field_idx.get(key)
                if i is None:
                    continue
                if key == "GT":
                    pv = ("s", v[1]) if isinstance(v, tuple) else None
                elif v is None:
                    pv = None
                elif isinstance(v, list):
                    pv = None if _wholly_missing(v) else ("a", v)
                elif isinstance(v, int):
                    pv = ("i", v)
                elif isinstance(v, float):
                    pv = ("f", v)
                else:
                    pv = ("s", v)
                if pv is not None:
                    cells[oi][i] = pv
            remaining -= 1
            if remaining == 0:
                break
        cols = []
        for i, tag in enumerate(self.format_fields):
            t = format_to_arrow_type(h.formats, tag)
            col = []
            for oi in range(ns):
                pv = cells[oi][i]
                if t == pa.utf8():
                    if pv is None or pv[0] == "a":
                        col.append(None)
                    elif pv[0] == "s":
                        col.append(pv[1])
                    elif pv[0] == "i":
                        col.append(str(pv[1]))
                    else:
                        raise VcfError("float into Utf8 FORMAT cell: not restated")
                elif t == pa.int32():
                    if pv is None:
                        col.append(None)
                    elif pv[0] == "i":
                        col.append(pv[1])
                    elif pv[0] == "a" and all(isinstance(x, int) or x is None for x in pv[1]):
                        col.append(next((x for x in pv[1] if x is not None), None))
                    else:
                        col.append(None)
                elif t == pa.float32():
                    if pv is None:
                        col.append(None)
                    elif pv[0] == "f":
                        col.append(pv[1])
                    elif pv[0] == "a" and all(isinstance(x, float) or x is None for x in pv[1]):
                        col.append(next((x for x in pv[1] if x is not None), None))
                    else:
                        col.append(None)
                elif pa.types.is_list(t):
                    et = t.value_type
                    if pv is None:
                        col.append(None)
                    elif pv[0] == "a":
                        col.append(pv[1])
                    elif (pv[0] == "i" and et == pa.int32()) or (pv[0] == "f" and et == pa.float32()) or (pv[0] == "s" and et == pa.utf8()):
                        col.append([pv[1]])
                    else:
                        col.append(None)
                else:
                    col.append(None)
            cols.append(col)
        return cols

    def _format_single_row(self, rec: Rec):
        """load_formats_single_pass for the single source sample -> per FORMAT field value."""
        h = self.header
        f = rec.f
        keys = f[8].split(":") if len(f) > 8 and f[8] not in ("", ".") else []
        samples = f[9:]
        nf = len(self.format_fields)
        vals = [None] * nf
        if not samples:
            return None  # no sample iterated: nothing appended (would misalign rows in the reference)
        field_idx = {t: i for i, t in enumerate(self.format_fields)}
        for key, v in parse_sample_values(keys, samples[0], h.formats, set(self.format_fields)):
            i = field_idx.get(key)
            if i is None:
                continue
            t = format_to_arrow_type(h.formats, self.format_fields[i])
            if key == "GT":
                vals[i] = v[1] if isinstance(v, tuple) else None
            elif v is None:
                vals[i] = None
            elif isinstance(v, list):
                if not pa.types.is_list(t) and len(v) != 1:
                    raise VcfError(f"FORMAT field '{key}' is declared scalar but the record contains {len(v)} values")
                if _wholly_missing(v):
                    vals[i] = None
                elif pa.types.is_list(t):
                    vals[i] = v
                else:
                    vals[i] = next((x for x in v if x is not None), None)
            else:
                vals[i] = [v] if pa.types.is_list(t) else v
        return vals

    # ---- execution ---------------------------------------------------------------------------------
    def _batches(self, recs, projection, batch_size, limit):
        fl = self._flags(projection)
        h = self.header
        multi = len(self.source_samples) > 1
        has_format = bool(self.format_fields) and bool(self.samples) and bool(self.source_samples) and (
            not multi or bool(self._sel_indices()))
        eff = choose_effective_batch_size(batch_size, fl["any_format"], len(self.format_fields), len(self.samples),
                                          len(self.source_samples))
        schema = self.projected_schema(projection)
        cols_idx = list(range(len(self.schema))) if projection is None else list(projection)
        rows = []
        total = 0
        for rec in recs:
            row = self._core_row(rec, fl)
            if fl["any_info"] and self.n_info:          # load_infos_single_pass returns at once without builders (:552-555)
                row["_info"] = self._info_row(rec, None if projection is None else {i - 8 for i in projection if 8 <= i < 8 + self.n_info})
            if has_format and fl["any_format"]:
                row["_fmt"] = self._format_multi_row(rec) if multi else self._format_single_row(rec)
            rows.append(row)
            total += 1
            if limit is not None and total >= limit:
                break
        batches = []
        for s in range(0, len(rows), eff):
            chunk = rows[s:s + eff]
            if not cols_idx:
                batches.append(pa.RecordBatch.from_struct_array(pa.array([{}] * len(chunk), type=pa.struct([]))))
                continue
            arrays = []
            for ci in cols_idx:
                fld = self.schema.field(ci)
                if ci < 8:
                    arrays.append(pa.array([r[fld.name] for r in chunk], type=fld.type))
                elif ci < 8 + self.n_info:
                    arrays.append(pa.array([r["_info"][ci - 8] for r in chunk], type=fld.type))
                elif multi:
                    kids = []
                    for k, tag in enumerate(self.format_fields):
                        kt = fld.type.field(k).type
                        kids.append(pa.array([r["_fmt"][k] for r in chunk], type=kt))
                    arrays.append(pa.StructArray.from_arrays(kids, fields=[fld.type.field(k) for k in range(len(kids))]))
                else:
                    k = ci - 8 - self.n_info
                    arrays.append(pa.array([r["_fmt"][k] for r in chunk], type=fld.type))
            batches.append(pa.RecordBatch.from_arrays(arrays, schema=schema))
        return schema, batches

    def _filter_fields(self, rec: Rec):
        start = u32(u32(rec.pos) - 1) if self.zero_based else u32(rec.pos)
        f = rec.f
        return {"chrom": rec.chrom, "start": start, "end": get_variant_end(rec, self.header.infos),
                "id": "" if f[2] == "." else f[2]}

    def execute(self, plan, partition: int = 0, batch_size: int = 8192):
        projection, limit = plan["projection"], plan["limit"]
        if plan["kind"] == "empty":
            return self.projected_schema(projection), []
        if plan["kind"] == "sequential":
            recs = (self._rec(t) for _, t in self._lines_from(self.data_start))
            return self._batches(recs, projection, batch_size, limit)
        residual = plan["residual"]
        if self.tbi is None:  # physical_exec.rs:2766-2768
            raise VcfError(f"Failed to open indexed VCF: {self.index_error}")

        def gen():
            for region in plan["assignments"][partition].regions:
                if region.unmapped_tail:
                    continue
                if region.start is not None and region.end is not None and region.end < region.start:
                    raise VcfError(f"Invalid region '{region.chrom}': end ({region.end}) is less than start ({region.start})")
                for rec in self._query(region):
                    if region.start is not None and u32(rec.pos) < region.start:   # physical_exec.rs:2875-2895: on the u32
                        continue
                    if region.end is not None and u32(rec.pos) > region.end:
                        continue
                    if residual and not evaluate_record_filters(self._filter_fields(rec), residual,
                                                                string_fields=("chrom", "id"), num_fields=("start", "end")):
                        continue
                    yield rec
        return self._batches(gen(), projection, batch_size, limit)


# =======================================================================================
# UDFs (udfs.rs)
# =======================================================================================
def list_avg(arr: pa.ListArray) -> pa.Array:
    """udfs.rs:67-110: sequential f64 accumulation over non-null elements; NULL list or no elements -> NULL."""
    out = []
    for i in range(len(arr)):
        if not arr[i].is_valid:
            out.append(None)
            continue
        s, c = 0.0, 0
        for v in arr[i].as_py():
            if v is not None:
                s += float(v)
                c += 1
        out.append(None if c == 0 else s / c)
    return pa.array(out, type=pa.float64())


def _list_cmp(arr: pa.ListArray, thr, op) -> pa.Array:
    out = []
    is_f = pa.types.is_floating(arr.type.value_type)
    if is_f:
        thr = float(np.float32(thr))
    for i in range(len(arr)):
        if not arr[i].is_valid:
            out.append(None)
            continue
        out.append([None if v is None else bool(op(v, thr)) for v in arr[i].as_py()])
    return pa.array(out, type=pa.list_(pa.field("item", pa.bool_(), True)))


def list_gte(arr, thr):  # udfs.rs:606-650
    return _list_cmp(arr, thr, lambda a, b: a >= b)


def list_lte(arr, thr):  # udfs.rs list_lte
    return _list_cmp(arr, thr, lambda a, b: a <= b)


def list_and(a: pa.ListArray, b: pa.ListArray) -> pa.Array:
    """udfs.rs:799-843: SQL three-valued AND element by element over min(len) elements."""
    out = []
    for i in range(len(a)):
        if not a[i].is_valid or not b[i].is_valid:
            out.append(None)
            continue
        row = []
        for l, r in zip(a[i].as_py(), b[i].as_py()):
            if l is None and r is None:
                row.append(None)
            elif l is None:
                row.append(None if r else False)
            elif r is None:
                row.append(None if l else False)
            else:
                row.append(l and r)
        out.append(row)
    return pa.array(out, type=pa.list_(pa.field("item", pa.bool_(), True)))


def vcf_set_gts(gt: pa.ListArray, mask: pa.ListArray, replacement: str = "./.") -> pa.Array:
    """udfs.rs:896-949: GT kept where the mask element is true, NULL or absent (and when the mask list is NULL);
    replaced where it is false; NULL GT elements stay NULL when kept; NULL GT list -> NULL list."""
    out = []
    for i in range(len(gt)):
        if not gt[i].is_valid:
            out.append(None)
            continue
        g = gt[i].as_py()
        m = mask[i].as_py() if mask[i].is_valid else None
        row = []
        for j, v in enumerate(g):
            keep = True if m is None else (j >= len(m) or m[j] is None or m[j])
            row.append(v if keep else replacement)
        out.append(row)
    return pa.array(out, type=pa.list_(pa.field("item", pa.utf8(), True)))


# ---- vcf_an / vcf_ac / vcf_af (bio-format-vcf/src/udfs.rs:113-154, 161-552) ---------------------------------------------------
def parse_gt_alleles(gt: str):
    """udfs.rs:117-142: None for an entirely missing genotype (".", "./.", ".|."); else one entry per piece between '/' and '|',
    None for "." and for anything Rust's `usize::from_str` rejects (empty, a sign other than one leading '+', a non-digit, overflow)."""
    gt = gt.strip()   # Rust's trim: Unicode White_Space, as Python's str.strip
    if gt in (".", "./.", ".|."):
        return None
    out = []
    for piece in gt.replace("|", "/").split("/"):
        a = piece.strip()
        if a == ".":
            out.append(None)
            continue
        digits = a[1:] if a[:1] == "+" else a
        if digits and all("0" <= c <= "9" for c in digits) and int(digits) < (1 << 64):
            out.append(int(digits))
        else:
            out.append(None)
    return out


def count_alt_alleles(alt: str) -> int:
    """udfs.rs:146-154"""
    alt = alt.strip()
    return 0 if alt in ("", ".") else len(alt.split("|"))


def _called(gts):
    for g in gts:
        if g is None:
            continue
        al = parse_gt_alleles(g)
        if al is not None:
            for a in al:
                if a is not None:
                    yield a


def vcf_an(gt: pa.Array) -> pa.Array:
    """udfs.rs:201-232"""
    return pa.array([None if not g.is_valid else sum(1 for _ in _called(g.as_py())) for g in gt], type=pa.int32())


def _ac_rows(gt: pa.Array, alt):
    for i, g in enumerate(gt):
        if not g.is_valid:
            yield None, 0
            continue
        called = list(_called(g.as_py()))
        vec_len = max(called, default=0)
        if alt is not None and alt[i].is_valid:
            vec_len = max(vec_len, count_alt_alleles(alt[i].as_py()))
        counts = [0] * vec_len
        for a in called:
            if 1 <= a <= vec_len:
                counts[a - 1] += 1
        yield counts, len(called)


def vcf_ac(gt: pa.Array, alt=None) -> pa.Array:
    """udfs.rs:283-381: the list is as long as max(number of ALT alleles when `alt` is given, largest called allele index)."""
    return pa.array([c for c, _ in _ac_rows(gt, alt)], type=pa.list_(pa.field("item", pa.int32(), True)))


def vcf_af(gt: pa.Array, alt=None) -> pa.Array:
    """udfs.rs:443-541: AC / AN; NULL elements when AN is 0."""
    out = []
    for c, an in _ac_rows(gt, alt):
        out.append(None if c is None else [None if an == 0 else x / an for x in c])
    return pa.array(out, type=pa.list_(pa.field("item", pa.float64(), True)))
