"""The reference's write -> read known answers (bio-format-bam/tests/write_test.rs), transcribed as data.

Each case is the Rust test's input batch and the literal values it asserts after reading the written file back.  The batch is
written by the product (`BamWriter.for_insert` = new_for_write + INSERT OVERWRITE: header, @SQ dictionary and coordinate
system from the Arrow schema alone) and read back three ways: by the HIP reader, by the independent CPU oracle
(oracle/bam_oracle.py: zlib + struct), and -- for the aux fields -- by parsing the record bytes directly.  These are the only
per-value vectors the reference holds for BAM columns, so they pin the writer and, through the oracle, the reader's name /
tag columns.

write_test.rs line numbers are given per case."""
import os
import struct
import zlib

import pyarrow as pa
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")

CORE = [("name", pa.string(), False), ("chrom", pa.string(), False), ("start", pa.uint32(), False), ("flags", pa.uint32(), False),
        ("cigar", pa.string(), False), ("mapping_quality", pa.uint32(), False), ("mate_chrom", pa.string(), True),
        ("mate_start", pa.uint32(), True), ("sequence", pa.string(), False), ("quality_scores", pa.string(), False),
        ("template_length", pa.int32(), False)]


def tag_field(name, typ, sam_type, description):
    return pa.field(name, typ, True, metadata={"bio.bam.tag.tag": name, "bio.bam.tag.type": sam_type, "bio.bam.tag.description": description})


def lst(t):
    return pa.list_(pa.field("item", t, True))


def core_cols(n, names=None, chroms=None, starts=None, flags=None, cigars=None, mapqs=None, seqs=None, quals=None):
    return [pa.array(names or ["read%d" % (i + 1) for i in range(n)], pa.string()),
            pa.array(chroms or ["chr1"] * n, pa.string()),
            pa.array(starts or [100 * (i + 1) for i in range(n)], pa.uint32()),
            pa.array(flags or [0] * n, pa.uint32()),
            pa.array(cigars or ["10M"] * n, pa.string()),
            pa.array(mapqs or [60] * n, pa.uint32()),
            pa.array([None] * n, pa.string()),
            pa.array([None] * n, pa.uint32()),
            pa.array(seqs or ["ACGTACGTAC"] * n, pa.string()),
            pa.array(quals or ["IIIIIIIIII"] * n, pa.string()),
            pa.array([0] * n, pa.int32())]


def make_batch(tag_fields, tag_cols, n, metadata=None, **core):
    fields = [pa.field(nm, t, nullable) for nm, t, nullable in CORE] + list(tag_fields)
    schema = pa.schema(fields, metadata=metadata)
    return pa.RecordBatch.from_arrays(core_cols(n, **core) + list(tag_cols), schema=schema)


def write(pkg, tmp_path, batch, sort_on_write=False, name="out.bam"):
    path = str(tmp_path / name)
    w = pkg.BamWriter.for_insert(path, batch.schema, sort_on_write=sort_on_write)
    w.write_records(batch)
    st = w.finish()
    assert st["n_records"] == batch.num_rows
    return path


def read_gpu(pkg, path, tag_fields, infer=True, sample=100, hints=None, columns=None):
    prov = pkg.BamTableProvider(path, None, True, tag_fields, False, infer, sample, hints, index_path="")
    schema = prov.schema()
    proj = None if columns is None else [schema.names.index(c) for c in columns]
    plan = prov.scan(projection=proj, target_partitions=1)
    batches = [b for p in range(plan.num_partitions()) for b in plan.execute(p, 8192)]
    return schema, pa.Table.from_batches(batches) if batches else None


def read_oracle(oracle, path, tag_fields, infer=True, sample=100, hints=None, columns=None):
    orc = oracle.BamOracle(path, True, tag_fields, False, infer, sample, hints, index_path=None)
    proj = None if columns is None else [orc.schema.names.index(c) for c in columns]
    _, batches = orc.execute_sequential(proj, 8192)
    return orc.schema, pa.Table.from_batches(batches) if batches else None


def both(pkg, oracle, path, tag_fields, **kw):
    sg, tg = read_gpu(pkg, path, tag_fields, **kw)
    so, to = read_oracle(oracle, path, tag_fields, **kw)
    assert sg.equals(so, check_metadata=False), (sg, so)
    assert tg.equals(to), "HIP reader and oracle disagree on the written file"
    return sg, tg


def aux_fields(path):
    """[{tag: (type, value)}] per record, parsed from the file's bytes with zlib + struct only."""
    data = open(path, "rb").read()
    u, o = b"", 0
    while o < len(data):
        bs = struct.unpack_from("<H", data, o + 16)[0] + 1
        u += zlib.decompress(data[o + 18:o + bs - 8], -15)
        o += bs
    assert u[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", u, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", u, p)[0]
    p += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", u, p)[0]
        p += 4 + ln + 4
    recs = []
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I", "f": "<f"}
    while p < len(u):
        bs = struct.unpack_from("<i", u, p)[0]
        r = u[p + 4:p + 4 + bs]
        p += 4 + bs
        l_name, n_cig, l_seq = r[8], struct.unpack_from("<H", r, 12)[0], struct.unpack_from("<i", r, 16)[0]
        q = 32 + l_name + 4 * n_cig + (l_seq + 1) // 2 + l_seq
        d = {}
        while q < len(r):
            tag, t = r[q:q + 2].decode(), chr(r[q + 2])
            q += 3
            if t in fmt:
                v = struct.unpack_from(fmt[t], r, q)[0]
                q += struct.calcsize(fmt[t])
            elif t == "A":
                v = chr(r[q]); q += 1
            elif t in "ZH":
                e = r.index(b"\0", q)
                v = r[q:e].decode(); q = e + 1
            elif t == "B":
                st, cnt = chr(r[q]), struct.unpack_from("<i", r, q + 1)[0]
                q += 5
                v = (st, [struct.unpack_from(fmt[st], r, q + k * struct.calcsize(fmt[st]))[0] for k in range(cnt)])
                q += cnt * struct.calcsize(fmt[st])
            d[tag] = (t, v)
        recs.append(d)
    return recs


def test_tags_round_trip(pkg, oracle, tmp_path):
    """write_test.rs:38-207: NM:i, MD:Z, AS:i on three reads."""
    tf = [tag_field("NM", pa.int32(), "i", "Edit distance"), tag_field("MD", pa.string(), "Z", "Mismatch positions"),
          tag_field("AS", pa.int32(), "i", "Alignment score")]
    b = make_batch(tf, [pa.array([2, 1, 0], pa.int32()), pa.array(["10", "5^A4", "10"]), pa.array([50, 45, 60], pa.int32())], 3,
                   chroms=["chr1", "chr1", "chr2"], flags=[0, 16, 0], seqs=["ACGTACGTAC", "ACGTACGTAC", "TTTTTTTTTT"])
    path = write(pkg, tmp_path, b)
    schema, t = both(pkg, oracle, path, ["NM", "MD", "AS"], columns=["name", "NM", "MD", "AS"])
    t = t.sort_by("name")
    assert t.num_rows == 3
    assert t["name"].to_pylist() == ["read1", "read2", "read3"]
    assert t["NM"].type == pa.int32() and t["NM"].to_pylist() == [2, 1, 0]
    assert t["MD"].type == pa.string() and t["MD"].to_pylist() == ["10", "5^A4", "10"]
    assert t["AS"].type == pa.int32() and t["AS"].to_pylist() == [50, 45, 60]
    # (beyond the Rust test) the other columns of the written rows: no @SQ in a schema without metadata -> chrom NULL
    _, full = both(pkg, oracle, path, ["NM", "MD", "AS"])
    full = full.sort_by("name")
    assert full["chrom"].to_pylist() == [None, None, None]
    assert full["start"].to_pylist() == [100, 200, 300] and full["flags"].to_pylist() == [0, 16, 0]
    assert full["cigar"].to_pylist() == ["10M"] * 3 and full["mapping_quality"].to_pylist() == [60] * 3
    assert full["sequence"].to_pylist() == ["ACGTACGTAC", "ACGTACGTAC", "TTTTTTTTTT"]
    assert full["quality_scores"].to_pylist() == ["IIIIIIIIII"] * 3
    assert full["template_length"].to_pylist() == [0, 0, 0] and full["mate_chrom"].to_pylist() == [None] * 3


def test_character_tags_round_trip(pkg, oracle, tmp_path):
    """write_test.rs:210-308: XT:A from a one-character string column, read back with the inferred schema (sample 10)."""
    b = make_batch([tag_field("XT", pa.string(), "A", "Type tag")], [pa.array(["U", "R"])], 2)
    path = write(pkg, tmp_path, b)
    _, t = both(pkg, oracle, path, ["XT"], sample=10, columns=["name", "XT"])
    t = t.sort_by("name")
    assert t["XT"].type == pa.string() and t["XT"].to_pylist() == ["U", "R"]
    assert [r["XT"] for r in aux_fields(path)] == [("A", "U"), ("A", "R")]


def test_integer_array_tags_round_trip(pkg, oracle, tmp_path):
    """write_test.rs:311-435: ZB:B:i from List<Int32>."""
    b = make_batch([tag_field("ZB", lst(pa.int32()), "B:i", "Base qualities")], [pa.array([[10, 20, 30], [5, 15, 25, 35]], lst(pa.int32()))], 2)
    path = write(pkg, tmp_path, b)
    _, t = both(pkg, oracle, path, ["ZB"], sample=10, columns=["name", "ZB"])
    t = t.sort_by("name")
    assert t["ZB"].type == lst(pa.int32())
    assert t["ZB"].to_pylist() == [[10, 20, 30], [5, 15, 25, 35]]
    assert [r["ZB"] for r in aux_fields(path)] == [("B", ("i", [10, 20, 30])), ("B", ("i", [5, 15, 25, 35]))]


def test_byte_array_tags_round_trip(pkg, oracle, tmp_path):
    """write_test.rs:438-560: ZC:B:C from List<UInt8>."""
    b = make_batch([tag_field("ZC", lst(pa.uint8()), "B:C", "Color space")], [pa.array([[1, 2, 3], [10, 20]], lst(pa.uint8()))], 2)
    path = write(pkg, tmp_path, b)
    _, t = both(pkg, oracle, path, ["ZC"], sample=10, columns=["name", "ZC"])
    t = t.sort_by("name")
    assert t["ZC"].type == lst(pa.uint8())
    assert t["ZC"].to_pylist() == [[1, 2, 3], [10, 20]]
    assert [r["ZC"] for r in aux_fields(path)] == [("B", ("C", [1, 2, 3])), ("B", ("C", [10, 20]))]


def test_float_tags_round_trip(pkg, oracle, tmp_path):
    """write_test.rs:563-663: XS:f from a Float64 column, read back as Float32 (tolerance 0.01 in the Rust test; the stored
    values are the f32 roundings of 45.5 and 38.2, asserted exactly here)."""
    b = make_batch([tag_field("XS", pa.float64(), "f", "Suboptimal alignment score")], [pa.array([45.5, 38.2], pa.float64())], 2)
    path = write(pkg, tmp_path, b)
    _, t = both(pkg, oracle, path, ["XS"], sample=10, columns=["name", "XS"])
    t = t.sort_by("name")
    assert t["XS"].type == pa.float32()
    got = t["XS"].to_pylist()
    assert abs(got[0] - 45.5) < 0.01 and abs(got[1] - 38.2) < 0.01
    assert got == [45.5, struct.unpack("<f", struct.pack("<f", 38.2))[0]]


def test_full_tag_type_round_trip(pkg, oracle, tmp_path):
    """write_test.rs:666-1073: every SAM type and array subtype from the Arrow storages the reference's test uses, read back
    with hints for the custom tags and the registry for ML / FZ / CG; an all-NULL array column (CG) writes nothing."""
    tf = [tag_field("ch", pa.string(), "A", "Custom character tag"), tag_field("sv", pa.string(), "Z", "Custom string tag"),
          tag_field("hx", pa.string(), "H", "Custom hex tag"), tag_field("de", pa.float64(), "f", "Custom float tag"),
          tag_field("ni", pa.int64(), "i", "Custom integer tag"), tag_field("ui", pa.uint64(), "I", "Custom unsigned integer tag"),
          tag_field("pc", lst(pa.int8()), "B:c", "Custom int8 array tag"), tag_field("pC", lst(pa.uint8()), "B:C", "Custom uint8 array tag"),
          tag_field("ps", lst(pa.int16()), "B:s", "Custom int16 array tag"), tag_field("pS", lst(pa.uint16()), "B:S", "Custom uint16 array tag"),
          tag_field("pa", lst(pa.int64()), "B:i", "Custom int32 array tag"), tag_field("pI", lst(pa.uint64()), "B:I", "Custom uint32 array tag"),
          tag_field("pf", lst(pa.float64()), "B:f", "Custom float array tag"), tag_field("ML", lst(pa.uint8()), "B:C", "Base modification probabilities"),
          tag_field("FZ", lst(pa.uint16()), "B:S", "Flow signal intensities"), tag_field("CG", lst(pa.uint64()), "B:I", "BAM-only CIGAR overflow tag")]
    cols = [pa.array(["A", "Z"]), pa.array(["alpha", "beta"]), pa.array(["0fa0", "beef"]), pa.array([1.5, 2.25], pa.float64()),
            pa.array([42, -7], pa.int64()), pa.array([3_000_000_000, 7], pa.uint64()),
            pa.array([[-1, 0, 5], [12, 34]], lst(pa.int8())), pa.array([[1, 2, 3], [250, 4]], lst(pa.uint8())),
            pa.array([[-30, 40], [300, 400]], lst(pa.int16())), pa.array([[10, 20], [50000]], lst(pa.uint16())),
            pa.array([[100000, 200000], [-5, 17]], lst(pa.int64())), pa.array([[100000, 200000], [4_000_000_000, 17]], lst(pa.uint64())),
            pa.array([[1.25, 2.5], [3.75]], lst(pa.float64())), pa.array([[4, 5, 6], [7, 8]], lst(pa.uint8())),
            pa.array([[10, 1000], [65000]], lst(pa.uint16())), pa.array([None, None], lst(pa.uint64()))]
    b = make_batch(tf, cols, 2, chroms=["chr1", "chr2"], starts=[100, 220], flags=[0, 16], cigars=["10M", "8M2S"], mapqs=[60, 42],
                   seqs=["ACGTACGTAC", "TTTTGGGGAA"], quals=["IIIIIIIIII", "JJJJJJJJJJ"])
    path = write(pkg, tmp_path, b)
    tags = [f.name for f in tf]
    hints = ["ch:A", "sv:Z", "hx:H", "de:f", "ni:i", "ui:I", "pc:B:c", "pC:B:C", "ps:B:s", "pS:B:S", "pa:B:i", "pI:B:I", "pf:B:f"]
    schema, t = both(pkg, oracle, path, tags, infer=False, hints=hints, columns=["name"] + tags)
    want_types = {"pc": lst(pa.int8()), "pC": lst(pa.uint8()), "ps": lst(pa.int16()), "pS": lst(pa.uint16()), "pa": lst(pa.int32()),
                  "pI": lst(pa.uint32()), "pf": lst(pa.float32()), "ML": lst(pa.uint8()), "FZ": lst(pa.uint16()), "CG": lst(pa.uint32())}
    for k, ty in want_types.items():
        assert schema.field(k).type == ty, k
    t = t.sort_by("name")
    assert t["ch"].to_pylist() == ["A", "Z"]
    assert t["sv"].to_pylist() == ["alpha", "beta"]
    assert t["hx"].to_pylist() == ["0FA0", "BEEF"]
    assert t["de"].to_pylist() == [1.5, 2.25]
    assert t["ni"].to_pylist() == [42, -7]
    assert t["ui"].to_pylist() == [3_000_000_000, 7]
    assert t["pc"].to_pylist() == [[-1, 0, 5], [12, 34]]
    assert t["pC"].to_pylist() == [[1, 2, 3], [250, 4]]
    assert t["ps"].to_pylist() == [[-30, 40], [300, 400]]
    assert t["pS"].to_pylist() == [[10, 20], [50000]]
    assert t["pa"].to_pylist() == [[100000, 200000], [-5, 17]]
    assert t["pI"].to_pylist() == [[100000, 200000], [4_000_000_000, 17]]
    assert t["pf"].to_pylist() == [[1.25, 2.5], [3.75]]
    assert t["ML"].to_pylist() == [[4, 5, 6], [7, 8]]
    assert t["FZ"].to_pylist() == [[10, 1000], [65000]]
    assert t["CG"].to_pylist() == [None, None]
    aux = aux_fields(path)
    assert "CG" not in aux[0] and "CG" not in aux[1]
    assert aux[0]["hx"] == ("H", "0FA0") and aux[1]["ui"] == ("I", 7) and aux[0]["ui"] == ("I", 3_000_000_000)
    assert aux[1]["ni"][1] == -7 and aux[0]["de"] == ("f", 1.5)
    assert list(aux[0].keys()) == [f.name for f in tf[:-1]]     # schema order, NULLs skipped


def test_read_add_write_read_round_trip(pkg, oracle, tmp_path):
    """write_test.rs:1076-1246: the first two rows of multi_chrom.bam (ORDER BY chrom, start), four tag columns added, written
    with the SOURCE schema's metadata (so the header carries its @SQ lines) and read back."""
    src = os.path.join(G, "multi_chrom.bam")
    _, full = read_gpu(pkg, src, None)
    cols = ["name", "chrom", "start", "end", "flags", "cigar", "mapping_quality", "mate_chrom", "mate_start", "sequence",
            "quality_scores", "template_length"]
    first2 = full.select(cols).sort_by([("chrom", "ascending"), ("start", "ascending")]).slice(0, 2).combine_chunks()
    src_schema = pkg.BamTableProvider(src, None, True, None, index_path="").schema()
    add = [tag_field("de", pa.float64(), "f", "Added float tag"), tag_field("sv", pa.string(), "Z", "Added string tag"),
           tag_field("pa", lst(pa.int64()), "B:i", "Added integer array tag"), tag_field("ML", lst(pa.uint8()), "B:C", "Added standard array tag")]
    fields = [src_schema.field(c) for c in cols] + add
    schema = pa.schema(fields, metadata=src_schema.metadata)
    arrays = [first2[c].chunk(0) if first2[c].num_chunks else pa.array([], first2[c].type) for c in cols]
    arrays += [pa.array([10.5, 11.25], pa.float64()), pa.array(["added-a", "added-b"]),
               pa.array([[1, 2, 3], [4, 5]], lst(pa.int64())), pa.array([[9, 8], [7, 6, 5]], lst(pa.uint8()))]
    b = pa.RecordBatch.from_arrays(arrays, schema=schema)
    path = write(pkg, tmp_path, b)
    tags = ["de", "sv", "pa", "ML"]
    _, t = both(pkg, oracle, path, tags, infer=False, hints=["de:f", "sv:Z", "pa:B:i"], columns=["name", "chrom", "start"] + tags)
    t = t.sort_by([("chrom", "ascending"), ("start", "ascending")])
    assert t.num_rows == 2
    assert t["de"].to_pylist() == [10.5, 11.25]
    assert t["sv"].to_pylist() == ["added-a", "added-b"]
    assert t["pa"].to_pylist()[0] == [1, 2, 3]
    assert t["ML"].to_pylist()[1] == [7, 6, 5]
    # (beyond the Rust test) every core column of the two rows survives, chrom included: the @SQ dictionary came from the schema
    _, back = both(pkg, oracle, path, tags, infer=False, hints=["de:f", "sv:Z", "pa:B:i"])
    back = back.sort_by([("chrom", "ascending"), ("start", "ascending")])
    for c in cols:
        assert back[c].to_pylist() == first2[c].to_pylist(), c
    hdr = oracle.BamOracle(path, index_path=None).hdr
    assert hdr.ref_names == oracle.BamOracle(src, index_path=None).hdr.ref_names
    assert hdr.text.split("\n")[0] == "@HD\tVN:1.6\tSO:unsorted" or hdr.text.split("\n")[0].startswith("@HD\tVN:")


def test_write_without_tags(pkg, oracle, tmp_path):
    """write_test.rs:1249-1307: one read, no tag columns; the file exists (and, beyond the Rust test, reads back)."""
    b = make_batch([], [], 1)
    path = write(pkg, tmp_path, b)
    assert os.path.exists(path)
    _, t = both(pkg, oracle, path, None)
    assert t["name"].to_pylist() == ["read1"] and t["sequence"].to_pylist() == ["ACGTACGTAC"] and t["start"].to_pylist() == [100]


def test_sort_on_write_coordinate_order(pkg, oracle, tmp_path):
    """write_test.rs:1310-1445: sort_on_write = true.  The sort is DataFusion's SortExec above the write plan (chrom, start
    ascending, nulls last); what the writer owns is the header's SO:coordinate and the @SQ dictionary from the schema's
    metadata.  The rows are handed over in the order SortExec produces."""
    md = {"bio.bam.reference_sequences": '[{"name":"chr1","length":249250621},{"name":"chr2","length":243199373}]'}
    b = make_batch([], [], 3, metadata=md, names=["read_c", "read_a", "read_b"], chroms=["chr2", "chr1", "chr1"], starts=[300, 100, 200],
                   seqs=["ACGTACGTAC", "ACGTACGTAC", "TTTTTTTTTT"])
    order = sorted(range(3), key=lambda i: (b["chrom"][i].as_py(), b["start"][i].as_py()))
    b = b.take(pa.array(order))
    path = write(pkg, tmp_path, b, sort_on_write=True)
    schema, t = both(pkg, oracle, path, None, columns=["name", "chrom", "start"])
    assert schema.metadata[b"bio.bam.sort_order"] == b"coordinate"
    assert t.num_rows == 3
    assert t["chrom"].to_pylist() == ["chr1", "chr1", "chr2"]
    assert t["start"].to_pylist() == [100, 200, 300]
    assert t["name"].to_pylist() == ["read_a", "read_b", "read_c"]


def test_sort_on_write_false_sets_unsorted(pkg, oracle, tmp_path):
    """write_test.rs:1448-1533: sort_on_write = false -> SO:unsorted."""
    md = {"bio.bam.reference_sequences": '[{"name":"chr1","length":249250621}]'}
    b = make_batch([], [], 1, metadata=md)
    path = write(pkg, tmp_path, b, sort_on_write=False)
    schema, t = both(pkg, oracle, path, None, columns=["name", "chrom", "start"])
    assert schema.metadata[b"bio.bam.sort_order"] == b"unsorted"
    assert t["chrom"].to_pylist() == ["chr1"] and t["start"].to_pylist() == [100]


def test_sam_path_is_refused(pkg, tmp_path):
    """BamCompressionType::from_path (writer.rs:27-43) picks the plain SAM writer for a .sam path; this library writes BGZF
    BAM only and says so."""
    b = make_batch([], [], 1)
    with pytest.raises(pkg.BioscanError, match=r"\.sam"):
        pkg.BamWriter.for_insert(str(tmp_path / "out.sam"), b.schema)
    with pytest.raises(pkg.BioscanError, match=r"\.sam"):
        pkg.BamWriter(str(tmp_path / "OUT.SAM"), "@HD\tVN:1.6\n", [], [])


def test_coordinate_system_comes_from_the_schema(pkg, oracle, tmp_path):
    """insert_into (table_provider.rs:1131-1135): the rows' coordinate system is the schema's bio.coordinate_system_zero_based,
    0-based when absent."""
    md = {"bio.bam.reference_sequences": '[{"name":"chr1","length":1000}]'}
    for zb, key in ((True, None), (True, "true"), (False, "false")):
        m = dict(md)
        if key:
            m["bio.coordinate_system_zero_based"] = key
        b = make_batch([], [], 1, metadata=m, starts=[100])
        path = write(pkg, tmp_path, b, name="cs_%s.bam" % key)
        pos = oracle.BamOracle(path, index_path=None)
        rec = next(iter(oracle.iter_records(pos.u, pos.hdr.first_record_offset)))
        assert rec.pos == (100 if zb else 99)      # BAM stores 0-based positions


def test_sliced_batches_and_columns_with_their_own_offsets(pkg, oracle, tmp_path):
    """Arrow arrays carry offsets: a sliced batch, and a batch whose columns were sliced separately, write the rows they
    show (the reference reads them through arrow-rs accessors, which apply the offsets)."""
    n = 40
    names = ["r%02d" % i for i in range(n)]
    starts = [10 * i for i in range(n)]
    nm = [None if i % 5 == 0 else i for i in range(n)]
    md = {"bio.bam.reference_sequences": '[{"name":"chr1","length":100000}]'}
    b = make_batch([tag_field("NM", pa.int32(), "i", "Edit distance")], [pa.array(nm, pa.int32())], n, metadata=md, names=names, starts=starts,
                   seqs=["ACGT" * (1 + i % 3) for i in range(n)], quals=["IIII" * (1 + i % 3) for i in range(n)],
                   cigars=["%dM" % (4 * (1 + i % 3)) for i in range(n)])
    sl = b.slice(7, 21)
    path = write(pkg, tmp_path, sl, name="sliced.bam")
    _, t = both(pkg, oracle, path, ["NM"])
    assert t["name"].to_pylist() == names[7:28] and t["start"].to_pylist() == starts[7:28] and t["NM"].to_pylist() == nm[7:28]
    assert t["sequence"].to_pylist() == ["ACGT" * (1 + i % 3) for i in range(7, 28)]
    # columns with different offsets: every column sliced out of a differently padded parent
    cols = []
    for k, c in enumerate(b.columns):
        pad = k % 4 + (3 if k == 2 else 0)
        parent = pa.concat_arrays([c.slice(0, pad), c]) if pad else c
        cols.append(parent.slice(pad, n))
    mixed = pa.RecordBatch.from_arrays(cols, schema=b.schema)
    assert len({c.offset for c in mixed.columns}) > 1
    path = write(pkg, tmp_path, mixed, name="mixed.bam")
    _, t = both(pkg, oracle, path, ["NM"])
    assert t["name"].to_pylist() == names and t["start"].to_pylist() == starts and t["NM"].to_pylist() == nm


def shim_writer_layout(batch, tag_fields, table_zero_based):
    """What shim/src/write.rs `writer_layout` hands to bioscan_bam_writer_*: the reference serialises exactly the columns
    NAMED in the table's resolved `tag_fields` (its schema's tag columns + the names given to `new_for_write`,
    table_provider.rs:1139-1154), in that order, SAM type = the field's `bio.bam.tag.type` or `Z` (`build_tag_data`,
    bio-format-core/src/sam_tag_io.rs:109-147), and takes the coordinate system from the TABLE schema (:1131-1135)."""
    sch = batch.schema
    listed = lambda n: n in tag_fields and len(n.encode()) == 2  # noqa: E731
    idx, fields = [], []
    for i, f in enumerate(sch):
        if listed(f.name):
            continue
        md = {k: v for k, v in (f.metadata or {}).items() if k != b"bio.bam.tag.tag"}
        idx.append(i)
        fields.append(f.with_metadata(md))
    for t in tag_fields:
        if len(t.encode()) == 2 and t in sch.names:
            f = sch.field(t)
            md = dict(f.metadata or {})
            md[b"bio.bam.tag.tag"] = t.encode()
            idx.append(sch.names.index(t))
            fields.append(f.with_metadata(md))
    md = dict(sch.metadata or {})
    md[b"bio.coordinate_system_zero_based"] = b"true" if table_zero_based else b"false"
    return pa.RecordBatch.from_arrays([batch.column(i) for i in idx], schema=pa.schema(fields, metadata=md))


def test_insert_writes_the_tables_tag_fields_and_coordinate_system(pkg, oracle, tmp_path):
    """an explicit tag without field metadata is written as type Z, a tag column of the INPUT that the table does
    not list is not written, the aux fields follow the table's tag order, and a 1-based table shifts POS whatever the input
    plan's schema says."""
    md = {"bio.bam.reference_sequences": '[{"name":"chr1","length":1000}]', "bio.coordinate_system_zero_based": "true"}
    xt = pa.field("XT", pa.string(), True)                                   # listed by new_for_write, no metadata at all
    nm = tag_field("NM", pa.int32(), "i", "edit distance")                   # carries tag metadata, the table does not list it
    as_ = tag_field("AS", pa.int32(), "i", "score")                          # the table schema's own tag column
    b = make_batch([xt, nm, as_], [pa.array(["abc", None], pa.string()), pa.array([1, 2], pa.int32()), pa.array([7, 8], pa.int32())],
                   2, metadata=md, starts=[100, 200])
    laid = shim_writer_layout(b, ["AS", "XT", "toolong"], table_zero_based=False)
    path = write(pkg, tmp_path, laid, name="layout.bam")
    recs = aux_fields(path)
    assert [list(r.items()) for r in recs] == [[("AS", ("i", 7)), ("XT", ("Z", "abc"))], [("AS", ("i", 8))]]
    o = oracle.BamOracle(path, index_path=None)
    assert [r.pos for r in oracle.iter_records(o.u, o.hdr.first_record_offset)] == [99, 199]   # 1-based rows -> 0-based BAM
