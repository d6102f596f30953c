"""Inline VCF inputs transcribed (as data) from the reference's tests, with the values those tests assert.
bio-format-vcf/tests/format_columns_test.rs:11-29, :441-459; info_missing_value_test.rs:10-19;
info_bare_key_test.rs:7-31; special_char_info_test.rs:7-11."""

SAMPLE_VCF_MULTI = (
    "##fileformat=VCFv4.3\n"
    "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Combined depth\">\n"
    "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
    "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Read depth\">\n"
    "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype quality\">\n"
    "##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"Allelic depths\">\n"
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSample1\tSample2\n"
    "chr1\t100\trs1\tA\tT\t60\tPASS\tDP=50\tGT:DP:GQ\t0/1:20:99\t1/1:30:95\n"
    "chr1\t200\trs2\tG\tC\t80\tPASS\tDP=60\tGT:DP:GQ\t0/0:25:99\t0/1:35:90\n"
    "chr2\t300\trs3\tC\tG\t70\tPASS\tDP=45\tGT:DP:GQ\t1|0:15:85\t./.:10:50\n"
    "chr2\t400\trs4\tT\tA\t50\tPASS\tDP=40\tGT:DP:GQ:AD\t0/1:18:92:10,8\t1/1:22:88:2,20\n"
)

SAMPLE_VCF_SINGLE = (
    "##fileformat=VCFv4.3\n"
    "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
    "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Read depth\">\n"
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tOnlySample\n"
    "chr1\t100\t.\tA\tT\t30\tPASS\t.\tGT:DP\t0/1:20\n"
    "chr1\t200\t.\tG\tC\t40\tPASS\t.\tGT:DP\t1|0:30\n"
)

SAMPLE_VCF_SINGLE_COLLISION = (
    "##fileformat=VCFv4.3\n"
    "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Combined depth across samples\">\n"
    "##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele frequency\">\n"
    "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
    "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Read depth\">\n"
    "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype quality\">\n"
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSampleA\n"
    "chr1\t100\trs1\tA\tT\t60\tPASS\tDP=50;AF=0.5\tGT:DP:GQ\t0/1:20:99\n"
    "chr1\t200\trs2\tG\tC\t80\tPASS\tDP=60;AF=0.3\tGT:DP:GQ\t1/1:30:95\n"
)

SAMPLE_VCF_SINGLE_DEEP_COLLISION = (
    "##fileformat=VCFv4.3\n"
    "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Combined depth\">\n"
    "##INFO=<ID=fmt_DP,Number=1,Type=Integer,Description=\"First fallback collision\">\n"
    "##INFO=<ID=format_DP,Number=1,Type=Integer,Description=\"Second fallback collision\">\n"
    "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"Sample depth\">\n"
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSampleA\n"
    "chr1\t100\trs1\tA\tT\t60\tPASS\tDP=50;fmt_DP=51;format_DP=52\tDP\t20\n"
)

VCF_WITH_MISSING_INFO_ARRAY = (
    "##fileformat=VCFv4.3\n"
    "##INFO=<ID=AD,Number=R,Type=Integer,Description=\"Allelic depths for ref and alt alleles\">\n"
    "##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele frequency\">\n"
    "##INFO=<ID=ALLELE_ID,Number=.,Type=String,Description=\"Allele identifiers\">\n"
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    "chr1\t100\trs1\tA\tT\t60\tPASS\tAD=.,15;AF=0.5;ALLELE_ID=.,alt1\n"
    "chr1\t200\trs2\tG\tC,T\t80\tPASS\tAD=10,.,5;AF=.,0.3;ALLELE_ID=ref2,.,alt2\n"
    "chr1\t300\trs3\tC\tT,A\t70\tPASS\tAD=5,.,10;AF=0.3,.;ALLELE_ID=ref3,alt3a,.\n"
    "chr1\t400\trs4\tT\tG\t90\tPASS\tAD=20,30;AF=0.6;ALLELE_ID=ref4,alt4\n"
)

VCF_WITH_BARE_NON_FLAG_INFO_KEYS = (
    "##fileformat=VCFv4.3\n"
    "##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Total depth\">\n"
    "##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele frequency\">\n"
    "##INFO=<ID=ALLELE_ID,Number=.,Type=String,Description=\"Allele identifiers\">\n"
    "##INFO=<ID=DB,Number=0,Type=Flag,Description=\"dbSNP membership\">\n"
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    "chr1\t100\trs1\tA\tT\t60\tPASS\tDP;AF=0.5;ALLELE_ID=alt1;DB\n"
    "chr1\t200\trs2\tG\tC\t80\tPASS\tDP=42;AF;ALLELE_ID=alt2\n"
    "chr1\t300\trs3\tC\tT\t70\tPASS\tDP=7;AF=0.2;ALLELE_ID\n"
    "chr1\t400\trs4\tT\tG\t90\tPASS\tDP=9;AF=0.3;ALLELE_ID=alt4;DB\n"
)

VCF_REALDATA_CHRX_EVIDENCE = (
    "##fileformat=VCFv4.2\n"
    "##contig=<ID=chrX,length=156040895>\n"
    "##INFO=<ID=AC,Number=A,Type=Integer,Description=\"Allele count for each ALT allele\">\n"
    "##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele frequency for each ALT allele\">\n"
    "##INFO=<ID=EVIDENCE,Number=.,Type=String,Description=\"Classes of random forest support\">\n"
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    "chrX\t1946351\tHGSV_249298\tA\t<DEL>\t.\t.\tAC=2;AF=0.998595;EVIDENCE\n"
)

VCF_WITH_INVALID_FLAG_VALUE = (
    "##fileformat=VCFv4.3\n"
    "##INFO=<ID=DB,Number=0,Type=Flag,Description=\"dbSNP membership\">\n"
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    "chr1\t100\trs1\tA\tT\t60\tPASS\tDB=unexpected_payload\n"
)

VCF_SPECIAL_INFO = (
    "##fileformat=VCFv4.3\n"
    "##INFO=<ID=HGMD-PUBLIC_20204,Number=0,Type=Flag,Description=\"Variants from HGMD-PUBLIC dataset December 2020\">\n"
    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    "chr1\t100\trs1\tA\tT\t60\tPASS\tHGMD-PUBLIC_20204\n"
)


def f32(x):
    import numpy as np
    return float(np.float32(x))


def check_reference_kats(make):
    """`make(text, **kw)` -> object with .read(projection_names=None) -> dict column -> python list (all rows).
    Asserts the values the reference's tests assert; shared by the oracle pin test and the GPU parity test."""
    # format_columns_test.rs:185-233
    t = make(SAMPLE_VCF_MULTI, info_fields=["DP"], format_fields=["GT", "DP"])
    g = t.read(["genotypes"])["genotypes"]
    assert g[0]["GT"] == ["0/1", "1/1"] and g[0]["DP"] == [20, 30]
    assert g[2]["GT"] == ["1|0", "./."] and g[2]["DP"] == [15, 10]
    # :257-330 sample subset keeps the requested order
    t = make(SAMPLE_VCF_MULTI, info_fields=["DP"], format_fields=["GT", "DP"], samples=["Sample2", "Sample1"])
    g = t.read(["genotypes"])["genotypes"]
    assert g[0]["GT"] == ["1/1", "0/1"] and g[0]["DP"] == [30, 20]
    # :332-376 one selected sample stays nested
    t = make(SAMPLE_VCF_MULTI, info_fields=["DP"], format_fields=["GT", "DP"], samples=["Sample2"])
    assert "genotypes" in t.column_names() and "GT" not in t.column_names()
    assert t.read(["genotypes"])["genotypes"][0]["GT"] == ["1/1"]
    # :400-441 missing requested samples are skipped
    t = make(SAMPLE_VCF_MULTI, info_fields=["DP"], format_fields=["GT"], samples=["MissingSample", "Sample1"])
    assert t.read(["genotypes"])["genotypes"][0]["GT"] == ["0/1"]
    # :235-255 single sample: top-level columns
    t = make(SAMPLE_VCF_SINGLE, format_fields=["GT", "DP"])
    r = t.read(["GT", "DP"])
    assert r["GT"] == ["0/1", "1|0"] and r["DP"] == [20, 30]
    assert "genotypes" not in t.column_names()
    # collisions (format_columns_test.rs:461-560, storage.rs:643-661)
    t = make(SAMPLE_VCF_SINGLE_COLLISION)
    assert t.column_names()[8:] == ["DP", "AF", "GT", "fmt_DP", "GQ"]
    r = t.read(["DP", "fmt_DP", "GT", "GQ"])
    assert r["DP"] == [50, 60] and r["fmt_DP"] == [20, 30] and r["GT"] == ["0/1", "1/1"] and r["GQ"] == [99, 95]
    t = make(SAMPLE_VCF_SINGLE_DEEP_COLLISION)
    assert "format_DP_2" in t.column_names()
    assert t.read(["format_DP_2"])["format_DP_2"] == [20]
    # info_missing_value_test.rs:36-154
    t = make(VCF_WITH_MISSING_INFO_ARRAY, info_fields=["AD", "AF", "ALLELE_ID"])
    r = t.read(["chrom", "AD", "AF", "ALLELE_ID"])
    assert r["chrom"] == ["chr1"] * 4
    assert r["AD"] == [[None, 15], [10, None, 5], [5, None, 10], [20, 30]]
    assert r["AF"] == [[f32(0.5)], [None, f32(0.3)], [f32(0.3), None], [f32(0.6)]]
    assert r["ALLELE_ID"] == [[None, "alt1"], ["ref2", None, "alt2"], ["ref3", "alt3a", None], ["ref4", "alt4"]]
    # info_bare_key_test.rs:77-243
    t = make(VCF_WITH_BARE_NON_FLAG_INFO_KEYS, info_fields=["DP", "AF", "ALLELE_ID", "DB"])
    r = t.read(["DP", "AF", "ALLELE_ID", "DB"])
    assert r["DP"] == [None, 42, 7, 9]
    assert r["AF"] == [[f32(0.5)], None, [f32(0.2)], [f32(0.3)]]
    assert r["ALLELE_ID"] == [["alt1"], ["alt2"], None, ["alt4"]]
    assert r["DB"] == [True, False, False, True]
    t = make(VCF_WITH_BARE_NON_FLAG_INFO_KEYS, info_fields=["AF"])
    r = t.read(["chrom", "AF"])
    assert r["AF"] == [[f32(0.5)], None, [f32(0.2)], [f32(0.3)]]
    t = make(VCF_REALDATA_CHRX_EVIDENCE, info_fields=["AC", "AF", "EVIDENCE"])
    r = t.read(["AC", "AF", "EVIDENCE", "qual", "filter", "alt", "end"])
    assert r["AC"] == [[2]] and r["AF"] == [[f32(0.998595)]] and r["EVIDENCE"] == [None]
    assert r["qual"] == [None] and r["filter"] == [""] and r["alt"] == ["<DEL>"] and r["end"] == [1946351]
    # special_char_info_test.rs
    t = make(VCF_SPECIAL_INFO, info_fields=["HGMD-PUBLIC_20204"])
    assert t.read(["HGMD-PUBLIC_20204"])["HGMD-PUBLIC_20204"] == [True]
    # info_bare_key_test.rs:245-262 explicit value for a Flag is an error
    t = make(VCF_WITH_INVALID_FLAG_VALUE, info_fields=["DB"])
    try:
        t.read(["DB"])
    except Exception as e:  # noqa: BLE001
        assert "invalid flag" in str(e) or "Error reading INFO field" in str(e)
    else:
        raise AssertionError("Flag INFO fields with explicit values must fail")
