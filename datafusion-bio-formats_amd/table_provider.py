"""ctypes binding of include/bioscan.h + the BamTableProvider / BamExec mirror."""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterator, Optional, Sequence

import pyarrow as pa

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libbioscan.so")
_lib = None


class BioscanError(RuntimeError):
    """DataFusionError::Execution analogue."""


class _ArrowSchema(C.Structure):
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_void_p), ("flags", C.c_int64),
                ("n_children", C.c_int64), ("children", C.c_void_p), ("dictionary", C.c_void_p),
                ("release", C.c_void_p), ("private_data", C.c_void_p)]


class _ArrowArray(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
                ("n_children", C.c_int64), ("buffers", C.c_void_p), ("children", C.c_void_p),
                ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class _Options(C.Structure):
    _fields_ = [("coordinate_system_zero_based", C.c_int32), ("tag_fields", C.POINTER(C.c_char_p)),
                ("n_tag_fields", C.c_int32), ("binary_cigar", C.c_int32), ("infer_tag_types", C.c_int32),
                ("infer_tag_sample_size", C.c_int32), ("tag_type_hints", C.POINTER(C.c_char_p)),
                ("n_tag_type_hints", C.c_int32), ("index_path", C.c_char_p), ("device_id", C.c_int32),
                ("chunk_members", C.c_int32)]


class _Literal(C.Structure):
    _fields_ = [("kind", C.c_int32), ("i", C.c_int64), ("f", C.c_double), ("s", C.c_char_p)]


class _Filter(C.Structure):
    _fields_ = [("column", C.c_char_p), ("op", C.c_int32), ("values", C.POINTER(_Literal)), ("n_values", C.c_int32)]


class _VcfOptions(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("coordinate_system_zero_based", C.c_int32),
                ("has_info_fields", C.c_int32), ("info_fields", C.POINTER(C.c_char_p)), ("n_info_fields", C.c_int32),
                ("has_format_fields", C.c_int32), ("format_fields", C.POINTER(C.c_char_p)), ("n_format_fields", C.c_int32),
                ("has_samples", C.c_int32), ("samples", C.POINTER(C.c_char_p)), ("n_samples", C.c_int32),
                ("index_path", C.c_char_p)]


class UdfStats(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("n_elements", C.c_uint64), ("count_a", C.c_uint64), ("count_b", C.c_uint64),
                ("sum", C.c_double), ("ms_kernel", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class ScanStats(C.Structure):
    _fields_ = [("n_blocks", C.c_uint64), ("compressed_bytes", C.c_uint64), ("inflated_bytes", C.c_uint64),
                ("arrow_bytes", C.c_uint64), ("n_records", C.c_uint64), ("n_rows", C.c_uint64),
                ("ms_h2d", C.c_double), ("ms_frame", C.c_double), ("ms_inflate", C.c_double), ("ms_chain", C.c_double),
                ("ms_extract", C.c_double), ("ms_total_gpu", C.c_double), ("ms_crc", C.c_double), ("ms_keys", C.c_double),
                ("ms_select", C.c_double), ("ms_wall", C.c_double), ("chain_iterations", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/bioscan.h declares (checked by the CPU test-suite)
EXPORTED_SYMBOLS = [
    "bioscan_bam_options_default", "bioscan_bam_open", "bioscan_schema", "bioscan_supports_filters_pushdown",
    "bioscan_scan", "bioscan_plan_num_partitions", "bioscan_plan_schema", "bioscan_plan_display",
    "bioscan_plan_partition_desc", "bioscan_execute", "bioscan_next", "bioscan_stream_close", "bioscan_plan_close",
    "bioscan_provider_close", "bioscan_last_error", "bioscan_provider_make_resident", "bioscan_provider_set_chunk_members", "bioscan_execute_device",
    "bioscan_bgzf_inflate", "bioscan_free", "bioscan_device_check",
    "bioscan_debug_balance_partitions", "bioscan_debug_plan_full_scan", "bioscan_fastq_open",
    "bioscan_vcf_options_default", "bioscan_vcf_open", "bioscan_udf_list_avg", "bioscan_udf_list_cmp", "bioscan_stream_list_udf", "bioscan_udf_list_and", "bioscan_udf_vcf_set_gts", "bioscan_udf_vcf_allele_stats",
    "bioscan_scan_devices", "bioscan_plan_partition_device", "bioscan_plan_make_resident", "bioscan_provider_resident_range",
    "bioscan_debug_shard_partitions", "bioscan_debug_extract_regions",
    "bioscan_bam_writer_open", "bioscan_bam_writer_open_schema", "bioscan_bam_header_from_schema", "bioscan_bam_writer_write", "bioscan_bam_writer_finish", "bioscan_bam_writer_close", "bioscan_bgzf_deflate",
]


def debug_extract_regions(filters, zero_based: bool = True) -> str:
    """The C++ planner's extract_genomic_regions on (column, op, value) filters (host only)."""
    filters = list(filters)
    arr, keep = _make_filters(filters)
    buf = C.create_string_buffer(1 << 14)
    load_library().bioscan_debug_extract_regions(arr, len(filters), 1 if zero_based else 0, buf, 1 << 14)
    return buf.value.decode()


def debug_shard_partitions(weights, world: int):
    """The C++ planner's partition -> device runs (bioscan_debug_shard_partitions; host only)."""
    n = len(weights)
    w = (C.c_uint64 * max(n, 1))(*weights)
    out = (C.c_int32 * max(n, 1))()
    runs = load_library().bioscan_debug_shard_partitions(w, n, world, out)
    return runs, [out[i] for i in range(n)]


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise BioscanError(f"{_LIB_PATH} is missing: build it with __graft_entry__.build() "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(_LIB_PATH)
    lib.bioscan_last_error.restype = C.c_char_p
    lib.bioscan_bam_open.argtypes = [C.c_char_p, C.POINTER(_Options), C.POINTER(C.c_void_p)]
    lib.bioscan_fastq_open.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.bioscan_schema.argtypes = [C.c_void_p, C.c_void_p]
    lib.bioscan_supports_filters_pushdown.argtypes = [C.c_void_p, C.POINTER(_Filter), C.c_int32, C.POINTER(C.c_int32)]
    lib.bioscan_scan.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.POINTER(_Filter), C.c_int32, C.c_int64,
                                 C.c_int32, C.POINTER(C.c_void_p)]
    lib.bioscan_plan_num_partitions.argtypes = [C.c_void_p]
    lib.bioscan_plan_schema.argtypes = [C.c_void_p, C.c_void_p]
    lib.bioscan_plan_display.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    lib.bioscan_plan_partition_desc.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32]
    lib.bioscan_execute.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.bioscan_execute_device.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(ScanStats), C.POINTER(C.c_void_p)]
    lib.bioscan_next.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    lib.bioscan_stream_close.argtypes = [C.c_void_p]
    lib.bioscan_plan_close.argtypes = [C.c_void_p]
    lib.bioscan_provider_close.argtypes = [C.c_void_p]
    lib.bioscan_provider_make_resident.argtypes = [C.c_void_p]
    lib.bioscan_provider_set_chunk_members.argtypes = [C.c_void_p, C.c_int32]
    lib.bioscan_bgzf_inflate.argtypes = [C.c_char_p, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_size_t), C.POINTER(C.c_double)]
    lib.bioscan_free.argtypes = [C.c_void_p]
    lib.bioscan_device_check.argtypes = [C.c_int32, C.c_char_p, C.c_int32]
    lib.bioscan_vcf_open.argtypes = [C.c_char_p, C.POINTER(_VcfOptions), C.POINTER(C.c_void_p)]
    lib.bioscan_udf_list_avg.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.bioscan_udf_list_cmp.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_void_p, C.c_void_p]
    lib.bioscan_udf_list_and.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.bioscan_udf_vcf_set_gts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.bioscan_udf_vcf_allele_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.bioscan_stream_list_udf.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_double, C.POINTER(UdfStats)]
    lib.bioscan_scan_devices.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.POINTER(_Filter), C.c_int32, C.c_int64,
                                         C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]
    lib.bioscan_plan_partition_device.argtypes = [C.c_void_p, C.c_int32]
    lib.bioscan_plan_make_resident.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]
    lib.bioscan_provider_resident_range.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.bioscan_bam_writer_open.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int32, C.c_int32,
                                            C.c_int32, C.POINTER(C.c_void_p)]
    lib.bioscan_bam_writer_open_schema.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.bioscan_bam_header_from_schema.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
    lib.bioscan_bam_writer_write.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bioscan_bam_writer_finish.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.bioscan_bam_writer_close.argtypes = [C.c_void_p]
    lib.bioscan_bgzf_deflate.argtypes = [C.c_char_p, C.c_size_t, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_double)]
    lib.bioscan_debug_extract_regions.argtypes = [C.POINTER(_Filter), C.c_int32, C.c_int32, C.c_char_p, C.c_int32]
    lib.bioscan_debug_shard_partitions.argtypes = [C.POINTER(C.c_uint64), C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise BioscanError(load_library().bioscan_last_error().decode("utf-8", "replace"))


def device_check(device_id: int = 0) -> str:
    lib = load_library()
    buf = C.create_string_buffer(256)
    _check(lib.bioscan_device_check(device_id, buf, 256))
    return buf.value.decode()


def bgzf_inflate(data: bytes, device_id: int = 0, check_crc: bool = True):
    """K1 alone: inflate every BGZF member of `data` on the GPU -> (bytes, kernel_ms)."""
    lib = load_library()
    out = C.c_void_p()
    n = C.c_size_t()
    ms = C.c_double()
    _check(lib.bioscan_bgzf_inflate(data, len(data), device_id, 1 if check_crc else 0, C.byref(out), C.byref(n), C.byref(ms)))
    try:
        return C.string_at(out, n.value), ms.value
    finally:
        lib.bioscan_free(out)


def debug_balance_partitions(estimates, target_partitions: int) -> str:
    """estimates: list of dicts {chrom,start,end,bytes,contig_len,unmapped,bins,leaf_span} -> plan text."""
    lib = load_library()
    n = len(estimates)
    if n == 0:
        return ""
    chroms = (C.c_char_p * n)(*[e["chrom"].encode() for e in estimates])
    u64 = C.c_uint64 * n
    rs = u64(*[e.get("start") or 0 for e in estimates])
    re_ = u64(*[e.get("end") or 0 for e in estimates])
    eb = u64(*[e["bytes"] for e in estimates])
    cl = u64(*[e.get("contig_len") or 0 for e in estimates])
    um = u64(*[e.get("unmapped") or 0 for e in estimates])
    bin_arrays = [(C.c_uint64 * max(len(e.get("bins") or []), 1))(*(e.get("bins") or [])) for e in estimates]
    bins = (C.POINTER(C.c_uint64) * n)(*[C.cast(a, C.POINTER(C.c_uint64)) for a in bin_arrays])
    nb = (C.c_int32 * n)(*[len(e.get("bins") or []) for e in estimates])
    leaf = max([e.get("leaf_span") or 0 for e in estimates])
    buf = C.create_string_buffer(1 << 20)
    lib.bioscan_debug_balance_partitions.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64),
                                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                                     C.POINTER(C.c_uint64), C.POINTER(C.POINTER(C.c_uint64)),
                                                     C.POINTER(C.c_int32), C.c_uint64, C.c_int32, C.c_char_p, C.c_int32]
    lib.bioscan_debug_balance_partitions(n, chroms, rs, re_, eb, cl, um, bins, nb, leaf, target_partitions, buf, 1 << 20)
    return buf.value.decode()


def debug_plan_full_scan(bai_path: str, ref_names, ref_lengths, target_partitions: int) -> str:
    lib = load_library()
    n = len(ref_names)
    names = (C.c_char_p * max(n, 1))(*[r.encode() for r in ref_names])
    lens = (C.c_int64 * max(n, 1))(*ref_lengths)
    buf = C.create_string_buffer(1 << 22)
    lib.bioscan_debug_plan_full_scan.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                                 C.c_int32, C.c_char_p, C.c_int32]
    rc = lib.bioscan_debug_plan_full_scan(bai_path.encode(), n, names, lens, target_partitions, buf, 1 << 22)
    if rc < 0:
        raise BioscanError("cannot read BAI " + bai_path)
    return buf.value.decode()


_OPS = {"=": 0, "!=": 1, "<": 2, "<=": 3, ">": 4, ">=": 5, "between": 6, "not between": 7, "in": 8, "not in": 9}


def _make_filters(filters):
    """(col, op, value) tuples -> C array (keeps python objects alive in the returned holder)."""
    keep = []
    arr = (_Filter * max(len(filters), 1))()
    for k, (col, op, val) in enumerate(filters):
        vals = [val] if op in ("=", "!=", "<", "<=", ">", ">=") else list(val)
        lits = (_Literal * max(len(vals), 1))()
        for j, v in enumerate(vals):
            if v is None:
                lits[j].kind = 0
            elif isinstance(v, bool):
                raise TypeError("bool literal")
            elif isinstance(v, int):
                lits[j].kind, lits[j].i = 1, v
            elif isinstance(v, float):
                lits[j].kind, lits[j].f = 2, v
            else:
                b = str(v).encode()
                keep.append(b)
                lits[j].kind, lits[j].s = 3, b
        cb = col.encode()
        keep += [cb, lits]
        arr[k].column, arr[k].op, arr[k].values, arr[k].n_values = cb, _OPS[op], lits, len(vals)
    return arr, keep


class BamTableProvider:
    """Mirror of BamTableProvider::new (bio-format-bam/src/table_provider.rs:381-390): same
    positional arguments (object_storage_options accepted and ignored: local files only)."""

    def __init__(self, file_path: str, object_storage_options=None, coordinate_system_zero_based: bool = True,
                 tag_fields: Optional[Sequence[str]] = None, binary_cigar: bool = False, infer_tag_types: bool = True,
                 infer_tag_sample_size: int = 100, tag_type_hints: Optional[Sequence[str]] = None,
                 device_id: int = 0, index_path: Optional[str] = None, chunk_members: int = 0):
        lib = load_library()
        o = _Options()
        lib.bioscan_bam_options_default(C.byref(o))
        o.coordinate_system_zero_based = 1 if coordinate_system_zero_based else 0
        self._keep = []
        if tag_fields is not None:
            tf = (C.c_char_p * max(len(tag_fields), 1))(*[t.encode() for t in tag_fields])
            o.tag_fields, o.n_tag_fields = tf, len(tag_fields)
            self._keep.append(tf)
        o.binary_cigar = 1 if binary_cigar else 0
        o.infer_tag_types = 1 if infer_tag_types else 0
        o.infer_tag_sample_size = infer_tag_sample_size
        if tag_type_hints:
            th = (C.c_char_p * len(tag_type_hints))(*[t.encode() for t in tag_type_hints])
            o.tag_type_hints, o.n_tag_type_hints = th, len(tag_type_hints)
            self._keep.append(th)
        if index_path is not None:
            o.index_path = index_path.encode()
        o.device_id = device_id
        o.chunk_members = chunk_members  # BGZF members per pipeline chunk of a stream (0 = default)
        self._h = C.c_void_p()
        _check(lib.bioscan_bam_open(file_path.encode(), C.byref(o), C.byref(self._h)))
        self.file_path = file_path

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            load_library().bioscan_provider_close(h)
            self._h = None

    def schema(self) -> pa.Schema:
        s = _ArrowSchema()
        _check(load_library().bioscan_schema(self._h, C.addressof(s)))
        return pa.Schema._import_from_c(C.addressof(s))

    def supports_filters_pushdown(self, filters):
        arr, keep = _make_filters(list(filters))
        out = (C.c_int32 * max(len(filters), 1))()
        _check(load_library().bioscan_supports_filters_pushdown(self._h, arr, len(filters), out))
        return ["Inexact" if out[i] else "Unsupported" for i in range(len(filters))]

    def make_resident(self):
        _check(load_library().bioscan_provider_make_resident(self._h))

    def resident_range(self, device_id: int = 0):
        """Compressed byte range [lo, hi) of the file resident on `device_id` ((0, 0): nothing)."""
        lo, hi = C.c_uint64(), C.c_uint64()
        _check(load_library().bioscan_provider_resident_range(self._h, device_id, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def scan(self, projection: Optional[Sequence[int]] = None, filters=(), limit: Optional[int] = None,
             target_partitions: int = 1, device_ids: Optional[Sequence[int]] = None) -> "BamExec":
        """TableProvider::scan.  device_ids: the GPUs of this node the plan's partitions are dealt to (contiguous runs in
        plan order, bioscan_scan_devices); None = the provider's device."""
        filters = list(filters)
        arr, keep = _make_filters(filters)
        if projection is None:
            proj, nproj = None, 0
        else:
            proj = (C.c_int32 * max(len(projection), 1))(*projection)
            nproj = len(projection)
        plan = C.c_void_p()
        lim = -1 if limit is None else limit
        if device_ids is None:
            _check(load_library().bioscan_scan(self._h, proj, nproj, arr, len(filters), lim, target_partitions, C.byref(plan)))
        else:
            dv = (C.c_int32 * max(len(device_ids), 1))(*device_ids)
            _check(load_library().bioscan_scan_devices(self._h, proj, nproj, arr, len(filters), lim, target_partitions, dv,
                                                       len(device_ids), C.byref(plan)))
        return BamExec(self, plan)


class BamExec:
    """Mirror of BamExec (bio-format-bam/src/physical_exec.rs:39-173)."""

    def __init__(self, provider: BamTableProvider, handle):
        self._provider = provider  # keeps the provider alive
        self._h = handle

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            load_library().bioscan_plan_close(h)
            self._h = None

    def name(self) -> str:
        return "BamExec"

    def num_partitions(self) -> int:
        return load_library().bioscan_plan_num_partitions(self._h)

    def schema(self) -> pa.Schema:
        s = _ArrowSchema()
        _check(load_library().bioscan_plan_schema(self._h, C.addressof(s)))
        return pa.Schema._import_from_c(C.addressof(s))

    def display(self) -> str:
        buf = C.create_string_buffer(4096)
        load_library().bioscan_plan_display(self._h, buf, 4096)
        return buf.value.decode()

    def partition_desc(self, partition: int) -> str:
        buf = C.create_string_buffer(1 << 16)
        load_library().bioscan_plan_partition_desc(self._h, partition, buf, 1 << 16)
        return buf.value.decode()

    def partition_device(self, partition: int) -> int:
        return load_library().bioscan_plan_partition_device(self._h, partition)

    def make_resident(self, partitions: Optional[Sequence[int]] = None):
        """Uploads now what the given partitions (None = all) will inflate, each to the device that executes it."""
        if partitions is None:
            _check(load_library().bioscan_plan_make_resident(self._h, None, 0))
        else:
            a = (C.c_int32 * max(len(partitions), 1))(*partitions)
            _check(load_library().bioscan_plan_make_resident(self._h, a, len(partitions)))

    def partition_estimated_bytes(self, partition: int) -> int:
        """PartitionAssignment.total_estimated_bytes (0 for the sequential single-partition plan)."""
        d = self.partition_desc(partition)
        return int(d.split("|", 1)[0]) if "|" in d else 0

    def execute(self, partition: int, batch_size: int = 8192) -> Iterator[pa.RecordBatch]:
        lib = load_library()
        st = C.c_void_p()
        _check(lib.bioscan_execute(self._h, partition, batch_size, C.byref(st)))
        schema = self.schema()
        try:
            while True:
                arr = _ArrowArray()
                has = C.c_int32()
                _check(lib.bioscan_next(st, C.addressof(arr), C.byref(has)))
                if not has.value:
                    break
                sch = _ArrowSchema()
                _check(lib.bioscan_plan_schema(self._h, C.addressof(sch)))
                if len(schema) == 0:
                    sa = pa.Array._import_from_c(C.addressof(arr), C.addressof(sch))
                    yield pa.RecordBatch.from_struct_array(sa).replace_schema_metadata(schema.metadata)
                else:
                    yield pa.RecordBatch._import_from_c(C.addressof(arr), C.addressof(sch))
        finally:
            lib.bioscan_stream_close(st)

    def execute_drain(self, partition: int, batch_size: int = 8192) -> dict:
        """Host stream, consumed the cheapest possible way: every exported batch is released at once (no pyarrow import).
        Measures what a consumer that keeps up sees: file resident in HBM -> Arrow buffers in host memory."""
        import time
        lib = load_library()
        st = C.c_void_p()
        t0 = time.perf_counter()
        _check(lib.bioscan_execute(self._h, partition, batch_size, C.byref(st)))
        rel_t = C.CFUNCTYPE(None, C.c_void_p)
        rows = batches = 0
        t_first = None
        try:
            while True:
                arr = _ArrowArray()
                has = C.c_int32()
                _check(lib.bioscan_next(st, C.addressof(arr), C.byref(has)))
                if not has.value:
                    break
                if t_first is None:
                    t_first = time.perf_counter() - t0
                rows += arr.length
                batches += 1
                rel_t(arr.release)(C.addressof(arr))
        finally:
            lib.bioscan_stream_close(st)
        return {"n_rows": rows, "n_batches": batches, "seconds": time.perf_counter() - t0, "seconds_to_first_batch": t_first or 0.0}

    def execute_device(self, partition: int, batch_size: int = 8192) -> dict:
        """Runs the whole partition on the GPU, leaves the Arrow buffers in HBM, returns stats."""
        lib = load_library()
        st = C.c_void_p()
        stats = ScanStats()
        _check(lib.bioscan_execute_device(self._h, partition, batch_size, C.byref(stats), C.byref(st)))
        lib.bioscan_stream_close(st)
        return stats.as_dict()


class FastqExec(BamExec):
    """Mirror of FastqExec (bio-format-fastq/src/physical_exec.rs:262-386)."""

    def name(self) -> str:
        return "FastqExec"


class FastqTableProvider:
    """Mirror of FastqTableProvider::new(file_path, object_storage_options)
    (bio-format-fastq/src/table_provider.rs:49-66); local files only."""

    def __init__(self, file_path: str, object_storage_options=None, device_id: int = 0, chunk_members: int = 0):
        lib = load_library()
        self._h = C.c_void_p()
        _check(lib.bioscan_fastq_open(file_path.encode(), device_id, C.byref(self._h)))
        self.file_path = file_path
        if chunk_members:
            self.set_chunk_members(chunk_members)

    def set_chunk_members(self, n: int):
        """BGZF members per pipeline chunk of the streams executed from now on (0 = default)."""
        _check(load_library().bioscan_provider_set_chunk_members(self._h, n))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            load_library().bioscan_provider_close(h)
            self._h = None

    def schema(self) -> pa.Schema:
        s = _ArrowSchema()
        _check(load_library().bioscan_schema(self._h, C.addressof(s)))
        return pa.Schema._import_from_c(C.addressof(s))

    def scan(self, projection: Optional[Sequence[int]] = None, filters=(), limit: Optional[int] = None,
             target_partitions: int = 1) -> FastqExec:
        if projection is None:
            proj, nproj = None, 0
        else:
            proj = (C.c_int32 * max(len(projection), 1))(*projection)
            nproj = len(projection)
        plan = C.c_void_p()
        _check(load_library().bioscan_scan(self._h, proj, nproj, None, 0, -1 if limit is None else limit, target_partitions,
                                           C.byref(plan)))
        return FastqExec(self, plan)


class VcfExec(BamExec):
    """Mirror of VcfExec (bio-format-vcf/src/physical_exec.rs:2539-2690)."""

    def name(self) -> str:
        return "VCFExec"

    def execute_device_stream(self, partition: int, batch_size: int = 8192) -> "VcfDeviceStream":
        """Runs the partition on the GPU and keeps its Arrow buffers in HBM for device-resident list UDFs."""
        lib = load_library()
        st = C.c_void_p()
        stats = ScanStats()
        _check(lib.bioscan_execute_device(self._h, partition, batch_size, C.byref(stats), C.byref(st)))
        return VcfDeviceStream(st, stats.as_dict())

    def execute_device_udf(self, partition: int, field: str, udf: str, threshold: float = 0.0, batch_size: int = 8192) -> dict:
        """Runs the partition on the GPU and applies a list UDF (`list_avg`, `list_gte`, `list_lte`) to
        `genotypes.<field>` without leaving HBM; returns scan stats + the UDF's checksums and kernel time."""
        lib = load_library()
        st = C.c_void_p()
        stats = ScanStats()
        _check(lib.bioscan_execute_device(self._h, partition, batch_size, C.byref(stats), C.byref(st)))
        try:
            us = UdfStats()
            code = {"list_avg": 0, "list_gte": 1, "list_lte": 2}[udf]
            _check(lib.bioscan_stream_list_udf(st, field.encode(), code, float(threshold), C.byref(us)))
        finally:
            lib.bioscan_stream_close(st)
        return {"scan": stats.as_dict(), "udf": us.as_dict()}


class VcfDeviceStream:
    """A partition result resident in HBM (bioscan_execute_device) + bioscan_stream_list_udf."""

    def __init__(self, handle, stats):
        self._h = handle
        self.stats = stats

    def list_udf(self, field: str, udf: str, threshold: float = 0.0) -> dict:
        us = UdfStats()
        code = {"list_avg": 0, "list_gte": 1, "list_lte": 2}[udf]
        _check(load_library().bioscan_stream_list_udf(self._h, field.encode(), code, float(threshold), C.byref(us)))
        return us.as_dict()

    def close(self):
        if self._h:
            load_library().bioscan_stream_close(self._h)
            self._h = None

    def __del__(self):
        self.close()


def _str_array(values):
    if values is None:
        return 0, None, 0, None
    enc = [v.encode() for v in values]
    arr = (C.c_char_p * max(len(enc), 1))(*enc)
    return 1, arr, len(enc), enc


class VcfTableProvider:
    """Mirror of VcfTableProvider::new / new_with_samples(file_path, info_fields, format_fields, samples_to_include,
    object_storage_options, coordinate_system_zero_based) (bio-format-vcf/src/table_provider.rs:752-814); local
    files only.  `None` selects every header INFO / FORMAT tag / sample, a list (possibly empty) is explicit."""

    def __init__(self, file_path: str, info_fields=None, format_fields=None, object_storage_options=None,
                 coordinate_system_zero_based: bool = True, samples_to_include=None, index_path: Optional[str] = None,
                 device_id: int = 0):
        lib = load_library()
        o = _VcfOptions()
        o.device_id = device_id
        o.coordinate_system_zero_based = 1 if coordinate_system_zero_based else 0
        o.has_info_fields, a1, o.n_info_fields, k1 = _str_array(info_fields)
        o.has_format_fields, a2, o.n_format_fields, k2 = _str_array(format_fields)
        o.has_samples, a3, o.n_samples, k3 = _str_array(samples_to_include)
        if a1 is not None:
            o.info_fields = a1
        if a2 is not None:
            o.format_fields = a2
        if a3 is not None:
            o.samples = a3
        o.index_path = index_path.encode() if index_path is not None else None
        self._h = C.c_void_p()
        _check(lib.bioscan_vcf_open(file_path.encode(), C.byref(o), C.byref(self._h)))
        self.file_path = file_path

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            load_library().bioscan_provider_close(h)
            self._h = None

    def schema(self) -> pa.Schema:
        s = _ArrowSchema()
        _check(load_library().bioscan_schema(self._h, C.addressof(s)))
        return pa.Schema._import_from_c(C.addressof(s))

    def supports_filters_pushdown(self, filters):
        arr, keep = _make_filters(filters)
        out = (C.c_int32 * max(len(filters), 1))()
        _check(load_library().bioscan_supports_filters_pushdown(self._h, arr, len(filters), out))
        return ["Inexact" if out[i] else "Unsupported" for i in range(len(filters))]

    def make_resident(self):
        _check(load_library().bioscan_provider_make_resident(self._h))

    def set_chunk_members(self, n: int):
        """BGZF members per pipeline chunk of the streams executed from now on (0 = default)."""
        _check(load_library().bioscan_provider_set_chunk_members(self._h, n))

    def scan(self, projection: Optional[Sequence[int]] = None, filters=(), limit: Optional[int] = None,
             target_partitions: int = 1) -> VcfExec:
        if projection is None:
            proj, nproj = None, 0
        else:
            proj = (C.c_int32 * max(len(projection), 1))(*projection)
            nproj = len(projection)
        arr, keep = _make_filters(filters)
        plan = C.c_void_p()
        _check(load_library().bioscan_scan(self._h, proj, nproj, arr, len(filters), -1 if limit is None else limit,
                                           target_partitions, C.byref(plan)))
        return VcfExec(self, plan)


def _export_array(arr: pa.Array):
    a, s = _ArrowArray(), _ArrowSchema()
    arr._export_to_c(C.addressof(a), C.addressof(s))
    return a, s


def list_avg(arr: pa.Array, device_id: int = 0) -> pa.Array:
    """`list_avg` UDF (bio-format-vcf/src/udfs.rs:24-115) on the GPU: List<Int32|Float32> -> Float64."""
    lib = load_library()
    a, s = _export_array(arr)
    oa, os_ = _ArrowArray(), _ArrowSchema()
    try:
        _check(lib.bioscan_udf_list_avg(C.addressof(a), C.addressof(s), device_id, C.addressof(oa), C.addressof(os_)))
    finally:
        pa.Array._import_from_c(C.addressof(a), C.addressof(s))  # releases the exported input
    return pa.Array._import_from_c(C.addressof(oa), C.addressof(os_))


def _list_cmp(arr: pa.Array, threshold, op: int, device_id: int) -> pa.Array:
    lib = load_library()
    a, s = _export_array(arr)
    oa, os_ = _ArrowArray(), _ArrowSchema()
    try:
        _check(lib.bioscan_udf_list_cmp(C.addressof(a), C.addressof(s), op, float(threshold), device_id, C.addressof(oa),
                                        C.addressof(os_)))
    finally:
        pa.Array._import_from_c(C.addressof(a), C.addressof(s))
    return pa.Array._import_from_c(C.addressof(oa), C.addressof(os_))


def list_gte(arr: pa.Array, threshold, device_id: int = 0) -> pa.Array:
    """`list_gte` UDF (udfs.rs:561-656)."""
    return _list_cmp(arr, threshold, 0, device_id)


def list_lte(arr: pa.Array, threshold, device_id: int = 0) -> pa.Array:
    """`list_lte` UDF (udfs.rs:663-760)."""
    return _list_cmp(arr, threshold, 1, device_id)


def list_and(a: pa.Array, b: pa.Array, device_id: int = 0) -> pa.Array:
    """`list_and` UDF (udfs.rs:765-850): element-wise SQL AND of two List<Boolean> arrays."""
    lib = load_library()
    aa, asch = _export_array(a)
    ba, bsch = _export_array(b)
    oa, os_ = _ArrowArray(), _ArrowSchema()
    try:
        _check(lib.bioscan_udf_list_and(C.addressof(aa), C.addressof(asch), C.addressof(ba), C.addressof(bsch), device_id,
                                        C.addressof(oa), C.addressof(os_)))
    finally:
        pa.Array._import_from_c(C.addressof(aa), C.addressof(asch))  # releases the exported inputs
        pa.Array._import_from_c(C.addressof(ba), C.addressof(bsch))
    return pa.Array._import_from_c(C.addressof(oa), C.addressof(os_))


def vcf_set_gts(gt: pa.Array, mask: pa.Array, replacement: str = "./.", device_id: int = 0) -> pa.Array:
    """`vcf_set_gts` UDF (udfs.rs:857-953)."""
    lib = load_library()
    ga, gsch = _export_array(gt)
    ma, msch = _export_array(mask)
    oa, os_ = _ArrowArray(), _ArrowSchema()
    try:
        _check(lib.bioscan_udf_vcf_set_gts(C.addressof(ga), C.addressof(gsch), C.addressof(ma), C.addressof(msch), replacement.encode(),
                                           device_id, C.addressof(oa), C.addressof(os_)))
    finally:
        pa.Array._import_from_c(C.addressof(ga), C.addressof(gsch))
        pa.Array._import_from_c(C.addressof(ma), C.addressof(msch))
    return pa.Array._import_from_c(C.addressof(oa), C.addressof(os_))


def bgzf_deflate(data: bytes, add_eof: bool = True, device_id: int = 0):
    """K-level write path: `data` as BGZF members (<= 65280 payload bytes each) compressed on the GPU -> (bytes, kernel ms)."""
    lib = load_library()
    out, n, ms = C.c_void_p(), C.c_size_t(), C.c_double()
    _check(lib.bioscan_bgzf_deflate(data, len(data), device_id, 1 if add_eof else 0, C.byref(out), C.byref(n), C.byref(ms)))
    try:
        return C.string_at(out, n.value), ms.value
    finally:
        lib.bioscan_free(out)


class BamWriter:
    """Mirror of BamLocalWriter (bio-format-bam/src/writer.rs:57-283): write_header at construction, write_records per
    RecordBatch, finish.  Serialisation, CRC32 and DEFLATE run on the GPU."""

    def __init__(self, path: str, header_text: str, ref_names: Sequence[str], ref_lengths: Sequence[int],
                 coordinate_system_zero_based: bool = True, device_id: int = 0):
        lib = load_library()
        names = (C.c_char_p * max(len(ref_names), 1))(*[n.encode() for n in ref_names])
        lens = (C.c_int64 * max(len(ref_lengths), 1))(*ref_lengths)
        self._h = C.c_void_p()
        _check(lib.bioscan_bam_writer_open(path.encode(), header_text.encode(), names, lens, len(ref_names),
                                           1 if coordinate_system_zero_based else 0, device_id, C.byref(self._h)))

    @classmethod
    def for_insert(cls, path: str, schema: pa.Schema, sort_on_write: bool = False, device_id: int = 0) -> "BamWriter":
        """The writer of an INSERT OVERWRITE (BamTableProvider::new_for_write + insert_into, table_provider.rs:639-662,
        1117-1178): header, @SQ dictionary and coordinate system all come from the Arrow schema's metadata."""
        lib = load_library()
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        sch = _ArrowSchema()
        schema._export_to_c(C.addressof(sch))
        try:
            _check(lib.bioscan_bam_writer_open_schema(path.encode(), C.addressof(sch), 1 if sort_on_write else 0, device_id, C.byref(self._h)))
        finally:
            if sch.release:
                C.CFUNCTYPE(None, C.c_void_p)(sch.release)(C.addressof(sch))
        return self

    def write_records(self, batch: pa.RecordBatch):
        arr, sch = _ArrowArray(), _ArrowSchema()
        pa.StructArray.from_arrays(batch.columns, fields=list(batch.schema))._export_to_c(C.addressof(arr), C.addressof(sch))
        try:
            _check(load_library().bioscan_bam_writer_write(self._h, C.addressof(arr), C.addressof(sch)))
        finally:
            rel_a = C.CFUNCTYPE(None, C.c_void_p)
            if arr.release:
                rel_a(arr.release)(C.addressof(arr))
            if sch.release:
                rel_a(sch.release)(C.addressof(sch))

    def finish(self) -> dict:
        r, m, b = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(load_library().bioscan_bam_writer_finish(self._h, C.byref(r), C.byref(m), C.byref(b)))
        return {"n_records": r.value, "n_members": m.value, "n_bytes": b.value}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            load_library().bioscan_bam_writer_close(h)
            self._h = None


def bam_header_from_schema(schema: pa.Schema, sort_on_write: Optional[bool] = None) -> str:
    """build_bam_header (bio-format-bam/src/header_builder.rs:42-195) + noodles' SAM header text, from schema metadata.
    sort_on_write None keeps the schema's own bio.bam.sort_order; True / False apply insert_into's override."""
    lib = load_library()
    sch = _ArrowSchema()
    schema._export_to_c(C.addressof(sch))
    out = C.c_void_p()
    try:
        _check(lib.bioscan_bam_header_from_schema(C.addressof(sch), -1 if sort_on_write is None else (1 if sort_on_write else 0), C.byref(out)))
        return C.string_at(out).decode()
    finally:
        if out:
            lib.bioscan_free(out)
        if sch.release:
            C.CFUNCTYPE(None, C.c_void_p)(sch.release)(C.addressof(sch))


def _allele_stats(gt: pa.Array, alt, which: int, device_id: int) -> pa.Array:
    lib = load_library()
    ga, gsch, aa, asch, oa, osch = _ArrowArray(), _ArrowSchema(), _ArrowArray(), _ArrowSchema(), _ArrowArray(), _ArrowSchema()
    gt._export_to_c(C.addressof(ga), C.addressof(gsch))
    if alt is not None:
        alt._export_to_c(C.addressof(aa), C.addressof(asch))
    rel = C.CFUNCTYPE(None, C.c_void_p)
    try:
        _check(lib.bioscan_udf_vcf_allele_stats(C.addressof(ga), C.addressof(gsch), C.addressof(aa) if alt is not None else None,
                                                C.addressof(asch) if alt is not None else None, which, device_id, C.addressof(oa), C.addressof(osch)))
        return pa.Array._import_from_c(C.addressof(oa), C.addressof(osch))
    finally:
        for st in (ga, gsch) + ((aa, asch) if alt is not None else ()):
            if st.release:
                rel(st.release)(C.addressof(st))


def vcf_an(gt: pa.Array, device_id: int = 0) -> pa.Array:
    """`vcf_an` UDF (udfs.rs:161-232): Int32 count of called alleles per row of a List<Utf8> GT column."""
    return _allele_stats(gt, None, 0, device_id)


def vcf_ac(gt: pa.Array, alt: Optional[pa.Array] = None, device_id: int = 0) -> pa.Array:
    """`vcf_ac` UDF (udfs.rs:238-392): List<Int32> count per ALT allele; `alt` = the pipe-separated ALT column (2-argument form)."""
    return _allele_stats(gt, alt, 1, device_id)


def vcf_af(gt: pa.Array, alt: Optional[pa.Array] = None, device_id: int = 0) -> pa.Array:
    """`vcf_af` UDF (udfs.rs:398-552): List<Float64> AC / AN per ALT allele (NULL elements when no allele is called)."""
    return _allele_stats(gt, alt, 2, device_id)
