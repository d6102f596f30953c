"""ctypes wrapper of oracle/bioscan_oracle.c (TEST INFRASTRUCTURE ONLY -- see that file's header)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import pyarrow as pa

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")


class _Col(C.Structure):
    _fields_ = [("data", C.c_void_p), ("offsets", C.c_void_p), ("valid", C.c_void_p), ("data_len", C.c_uint64)]


class _Result(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("n_blocks", C.c_uint64), ("compressed_bytes", C.c_uint64),
                ("inflated_bytes", C.c_uint64), ("cols", _Col * 20), ("n_cols", C.c_int),
                ("seconds_inflate", C.c_double), ("seconds_chain", C.c_double), ("seconds_columns", C.c_double),
                ("seconds_total", C.c_double), ("threads", C.c_int), ("used_libdeflate", C.c_int), ("error", C.c_char * 256)]


_lib = None
CORE = [("name", "s", False), ("chrom", "s", True), ("start", "u", True), ("end", "u", True), ("flags", "u", False),
        ("cigar", "s", False), ("mapping_quality", "u", False), ("mate_chrom", "s", True), ("mate_start", "u", True),
        ("sequence", "s", False), ("quality_scores", "s", False), ("template_length", "i", False)]


def lib():
    global _lib
    if _lib is None:
        import subprocess
        subprocess.check_call(["make", "-s", "-C", _HERE])  # no-op when _build/liboracle.so is newer than the source
        _lib = C.CDLL(_LIB)
        _lib.oracle_bam_scan_mem.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_char_p,
                                             C.POINTER(C.c_int), C.c_int, C.POINTER(_Result)]
        _lib.oracle_free.argtypes = [C.POINTER(_Result)]
    return _lib


def scan(data: bytes, zero_based=True, threads=1, max_blocks=0, tags=(), tag_kinds=(), build_columns=True, to_arrow=True):
    """Returns (stats dict, {column name: pyarrow array with 64-bit offsets}) for a sequential full scan."""
    res = _Result()
    tg = "".join(tags).encode()
    kinds = (C.c_int * max(len(tags), 1))(*[0 if k == "i" else 3 for k in tag_kinds])
    rc = lib().oracle_bam_scan_mem(data, len(data), 1 if zero_based else 0, threads, max_blocks, len(tags), tg, kinds,
                                   1 if build_columns else 0, C.byref(res))
    if rc:
        raise RuntimeError(res.error.decode())
    stats = {k: getattr(res, k) for k in ("n_rows", "n_blocks", "compressed_bytes", "inflated_bytes", "seconds_inflate",
                                          "seconds_chain", "seconds_columns", "seconds_total", "threads", "used_libdeflate")}
    cols = {}
    if build_columns and to_arrow:
        n = res.n_rows
        spec = CORE + [(t, "i" if k == "i" else "s", True) for t, k in zip(tags, tag_kinds)]
        for c, (name, kind, nullable) in enumerate(spec):
            col = res.cols[c]
            vb = None
            if nullable and col.valid:
                vb = pa.py_buffer(C.string_at(col.valid, (n + 7) // 8))
            if kind == "s":
                ob = pa.py_buffer(C.string_at(col.offsets, (n + 1) * 8))
                db = pa.py_buffer(C.string_at(col.data, col.data_len))
                cols[name] = pa.Array.from_buffers(pa.large_utf8(), n, [vb, ob, db])
            else:
                db = pa.py_buffer(C.string_at(col.data, n * 4))
                cols[name] = pa.Array.from_buffers(pa.uint32() if kind == "u" else pa.int32(), n, [vb, db])
    lib().oracle_free(C.byref(res))
    return stats, cols


class _StreamResult(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("n_batches", C.c_uint64), ("n_blocks", C.c_uint64), ("compressed_bytes", C.c_uint64),
                ("inflated_bytes", C.c_uint64), ("arrow_bytes", C.c_uint64), ("col_bytes", C.c_uint64 * 12), ("col_sum", C.c_uint64 * 12),
                ("n_valid", C.c_uint64 * 12), ("seconds_total", C.c_double), ("seconds_inflate_avg", C.c_double),
                ("seconds_inflate_max", C.c_double), ("seconds_build_avg", C.c_double), ("seconds_build_max", C.c_double),
                ("threads", C.c_int), ("used_libdeflate", C.c_int), ("error", C.c_char * 256)]


def stream_plan(data: bytes, threads: int, max_blocks: int = 0):
    """The "index" of the streaming baseline: record-aligned virtual offsets of `threads` partitions over the first
    max_blocks members (untimed).  Returns (start_coff[], start_within[], end_coff, n_blocks)."""
    L = lib()
    L.oracle_bam_stream_plan.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64), C.c_char_p, C.c_int]
    sc, sw = (C.c_uint64 * threads)(), (C.c_uint64 * threads)()
    nb = C.c_uint64()
    err = C.create_string_buffer(256)
    if L.oracle_bam_stream_plan(data, len(data), max_blocks, threads, sc, sw, C.byref(nb), err, 256):
        raise RuntimeError(err.value.decode())
    # end of the last complete member of the sample
    import struct
    o = 0
    for _ in range(nb.value):
        o += struct.unpack_from("<H", data, o + 16)[0] + 1
    return list(sc), list(sw), o, nb.value


def stream_plan_bai(data: bytes, bai_bytes: bytes, threads: int, max_blocks: int = 0):
    """The same plan from the file's BAI, the way the reference gets its partitions' record-aligned starts (every chunk
    begin and every linear-index entry of a BAI is the virtual offset of a record start): partition t begins at the first
    indexed record start in or behind member n_blocks * t / threads.  Costs no inflate."""
    import bisect
    import struct
    import bam_oracle
    coffs, o = [], 0
    while o + 28 <= len(data) and (max_blocks == 0 or len(coffs) < max_blocks):
        bs = struct.unpack_from("<H", data, o + 16)[0] + 1
        if o + bs > len(data):
            break
        coffs.append(o)
        o += bs
    end_coff, nb = o, len(coffs)
    bai = bam_oracle.parse_bai(bai_bytes)
    vo = set()
    for r in bai.refs:
        for chunks in r.bins.values():
            vo.update(c[0] for c in chunks)
        vo.update(v for v in r.intervals if v)
    vo = sorted(v for v in vo if (v >> 16) < end_coff)
    sc, sw = [], []
    for t in range(threads):
        lo = coffs[nb * t // threads] << 16 if t else 0
        k = bisect.bisect_left(vo, lo)
        v = vo[k] if k < len(vo) else (end_coff << 16)
        sc.append(v >> 16)
        sw.append(v & 0xFFFF)
    return sc, sw, end_coff, nb


def stream_scan(data: bytes, plan, zero_based=True, batch_rows=8192) -> dict:
    """oracle_bam_scan_stream: one thread per partition of `plan`, member by member, per-batch builders (the reference's
    executor shape).  Returns counts, order-independent column checksums and the per-thread time split."""
    L = lib()
    sc, sw, end_coff, _ = plan
    T = len(sc)
    L.oracle_bam_scan_stream.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                         C.c_uint64, C.c_uint32, C.POINTER(_StreamResult)]
    res = _StreamResult()
    rc = L.oracle_bam_scan_stream(data, end_coff, 1 if zero_based else 0, T, (C.c_uint64 * T)(*sc), (C.c_uint64 * T)(*sw), end_coff,
                                  batch_rows, C.byref(res))
    if rc:
        raise RuntimeError(res.error.decode())
    out = {k: getattr(res, k) for k, _ in _StreamResult._fields_ if k not in ("col_bytes", "col_sum", "n_valid", "error")}
    names = [n for n, _, _ in CORE]
    out["col_bytes"] = dict(zip(names, list(res.col_bytes)))
    out["col_sum"] = dict(zip(names, list(res.col_sum)))
    out["n_valid"] = dict(zip(names, list(res.n_valid)))
    return out


class _VcfResult(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("n_blocks", C.c_uint64), ("compressed_bytes", C.c_uint64), ("inflated_bytes", C.c_uint64),
                ("sum_start", C.c_uint64), ("sum_end", C.c_uint64), ("core_str_bytes", C.c_uint64), ("n_qual_valid", C.c_uint64),
                ("sum_qual", C.c_double),
                ("info_int_sum", C.c_uint64), ("info_valid", C.c_uint64), ("info_float_valid", C.c_uint64),
                ("info_str_bytes", C.c_uint64), ("info_flag_true", C.c_uint64), ("info_list_elems", C.c_uint64),
                ("info_float_sum", C.c_double),
                ("cells", C.c_uint64), ("gt_bytes", C.c_uint64), ("gt_valid", C.c_uint64), ("gq_sum", C.c_uint64),
                ("gq_valid", C.c_uint64), ("dp_sum", C.c_uint64), ("dp_valid", C.c_uint64),
                ("avg_gq_sum", C.c_double), ("avg_dp_sum", C.c_double),
                ("avg_gq_valid", C.c_uint64), ("avg_dp_valid", C.c_uint64), ("gq_gte_true", C.c_uint64),
                ("dp_gte_true", C.c_uint64), ("dp_lte_true", C.c_uint64),
                ("seconds_inflate", C.c_double), ("seconds_parse", C.c_double), ("seconds_total", C.c_double),
                ("threads", C.c_int), ("used_libdeflate", C.c_int), ("error", C.c_char * 256)]


VCF_KINDS = {"int": 0, "float": 1, "flag": 2, "string": 3, "list_int": 4, "list_float": 5, "list_string": 6}


def vcf_scan(data: bytes, info=(), n_samples=0, zero_based=True, threads=1, b0=0, b1=0, x0=0, x1=0) -> dict:
    """C restatement of the VCF scan over BGZF members [b0, b1) (b1 = 0: all), text range [x0, x1) relative to the
    first decoded byte of member b0.  `info` = [(name, kind)] with kind in VCF_KINDS.  Returns counts + column checksums."""
    L = lib()
    L.oracle_vcf_scan_mem.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                                      C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.POINTER(_VcfResult)]
    names = (C.c_char_p * max(len(info), 1))(*[n.encode() for n, _ in info])
    kinds = (C.c_int * max(len(info), 1))(*[VCF_KINDS[k] for _, k in info])
    res = _VcfResult()
    rc = L.oracle_vcf_scan_mem(data, len(data), b0, b1, x0, x1, 1 if zero_based else 0, threads, len(info), names, kinds,
                               n_samples, C.byref(res))
    if rc:
        raise RuntimeError(res.error.decode())
    return {k: getattr(res, k) for k, _ in _VcfResult._fields_ if k != "error"}


class _FastqResult(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("n_blocks", C.c_uint64), ("compressed_bytes", C.c_uint64), ("inflated_bytes", C.c_uint64),
                ("name_bytes", C.c_uint64), ("desc_bytes", C.c_uint64), ("seq_bytes", C.c_uint64), ("qual_bytes", C.c_uint64),
                ("desc_null", C.c_uint64), ("byte_sum", C.c_uint64),
                ("seconds_inflate", C.c_double), ("seconds_parse", C.c_double), ("seconds_total", C.c_double),
                ("threads", C.c_int), ("used_libdeflate", C.c_int), ("error", C.c_char * 256)]


def fastq_scan(data: bytes, threads=1, max_blocks=0) -> dict:
    """C restatement of the BGZF-FASTQ scan over the first max_blocks members (0 = all): four Utf8 columns are built
    per thread; returns row count, per-column byte totals, NULL descriptions and a weighted byte checksum."""
    L = lib()
    L.oracle_fastq_scan_mem.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(_FastqResult)]
    res = _FastqResult()
    rc = L.oracle_fastq_scan_mem(data, len(data), max_blocks, threads, C.byref(res))
    if rc:
        raise RuntimeError(res.error.decode())
    return {k: getattr(res, k) for k, _ in _FastqResult._fields_ if k != "error"}
