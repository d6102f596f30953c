"""SAM header from Arrow schema metadata (bio-format-bam/src/header_builder.rs:42-195) -- host code of the write path, no
GPU needed: the library's bioscan_bam_header_from_schema against the oracle restatement and against the reference's own
unit tests (header_builder.rs:253-400, transcribed as data), and a round trip through the reader's metadata extraction:
the header of every committed BAM fixture, rebuilt from the metadata the oracle extracts from it, has the same @HD / @SQ /
@RG / @PG / @CO content."""
import json
import os
import random
import sys

import pyarrow as pa
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _schema(md):
    return pa.schema([pa.field("name", pa.string()), pa.field("chrom", pa.string()), pa.field("start", pa.uint32())], metadata=md or None)


def _lines(text):
    assert text.endswith("\n")
    return text[:-1].split("\n")


def test_reference_unit_tests(pkg, oracle):
    # test_build_bam_header_basic: defaults
    t = pkg.bam_header_from_schema(_schema({}))
    assert t == "@HD\tVN:1.6\n" == oracle.build_bam_header({})
    # test_build_bam_header_with_metadata: version, sort order, two reference sequences
    md = {"bio.bam.file_format_version": "1.6", "bio.bam.sort_order": "coordinate",
          "bio.bam.reference_sequences": json.dumps([{"name": "chr1", "length": 249250621}, {"name": "chr2", "length": 242193529}])}
    t = pkg.bam_header_from_schema(_schema(md))
    assert _lines(t) == ["@HD\tVN:1.6\tSO:coordinate", "@SQ\tSN:chr1\tLN:249250621", "@SQ\tSN:chr2\tLN:242193529"]
    assert t == oracle.build_bam_header(md)
    # test_build_bam_header_with_read_groups
    md = {"bio.bam.read_groups": json.dumps([{"id": "RG1", "sample": "SAMPLE1", "platform": "ILLUMINA", "library": "LIB1",
                                               "description": "Test read group"}])}
    t = pkg.bam_header_from_schema(_schema(md))
    assert _lines(t)[1:] == ["@RG\tID:RG1\tSM:SAMPLE1\tPL:ILLUMINA\tLB:LIB1\tDS:Test read group"]
    assert t == oracle.build_bam_header(md)
    # test_build_bam_header_with_programs
    md = {"bio.bam.program_info": json.dumps([{"id": "bwa", "name": "bwa", "version": "0.7.17", "command_line": "bwa mem ref.fa reads.fq"}])}
    t = pkg.bam_header_from_schema(_schema(md))
    assert _lines(t)[1:] == ["@PG\tID:bwa\tPN:bwa\tVN:0.7.17\tCL:bwa mem ref.fa reads.fq"]
    assert t == oracle.build_bam_header(md)
    # test_build_bam_header_with_comments
    md = {"bio.bam.comments": json.dumps(["This is a test", "Another comment"])}
    t = pkg.bam_header_from_schema(_schema(md))
    assert _lines(t)[1:] == ["@CO\tThis is a test", "@CO\tAnother comment"]
    assert t == oracle.build_bam_header(md)


def test_sort_order_override_of_insert_into(pkg, oracle):
    # table_provider.rs:1156-1164: sort_on_write decides @HD SO whatever the schema's metadata says
    md = {"bio.bam.sort_order": "queryname", "bio.bam.group_order": "query", "bio.bam.subsort_order": "coordinate:queryname"}
    assert _lines(pkg.bam_header_from_schema(_schema(md)))[0] == "@HD\tVN:1.6\tSO:queryname\tGO:query\tSS:coordinate:queryname"
    assert _lines(pkg.bam_header_from_schema(_schema(md), True))[0] == "@HD\tVN:1.6\tSO:coordinate\tGO:query\tSS:coordinate:queryname"
    assert _lines(pkg.bam_header_from_schema(_schema(md), False))[0] == "@HD\tVN:1.6\tSO:unsorted\tGO:query\tSS:coordinate:queryname"
    for sow in (None, True, False):
        assert pkg.bam_header_from_schema(_schema(md), sow) == oracle.build_bam_header(md, sow)


def test_rules_and_malformed_metadata(pkg, oracle):
    cases = [
        {"bio.bam.file_format_version": "banana"},                       # does not parse as major.minor -> 1.6
        {"bio.bam.file_format_version": "1.4"},
        {"bio.bam.file_format_version": "01.05"},
        {"bio.bam.reference_sequences": "not json"},                     # from_json_string -> None: no @SQ
        {"bio.bam.reference_sequences": json.dumps([{"name": "a"}])},    # a required field is missing: the whole list is dropped
        {"bio.bam.reference_sequences": json.dumps([{"name": "c", "length": 10, "other_fields": {"M5": "x", "AS": "y", "zz": "dropped", "UR": "file:/r.fa"}}])},
        {"bio.bam.read_groups": json.dumps([{"id": "g", "other_fields": {"PU": "unit", "CN": "centre", "SM": "ignored here", "BC": "ACGT"}},
                                             {"id": "h", "sample": None, "platform": "ONT"}])},
        {"bio.bam.program_info": json.dumps([{"id": "p", "other_fields": {"PP": "prev", "DS": "desc", "XX": "no"}}])},
        {"bio.bam.comments": json.dumps(["tab\tinside", "quote \" and \\ backslash", "unicode é中"])},
        {"bio.bam.comments": json.dumps([1, 2])},                        # not strings: dropped
    ]
    for md in cases:
        assert pkg.bam_header_from_schema(_schema(md)) == oracle.build_bam_header(md), md
    assert "\tAS:y\tM5:x\tUR:file:/r.fa" in pkg.bam_header_from_schema(_schema(cases[5]))
    with pytest.raises(pkg.BioscanError, match="Reference sequence length cannot be zero"):
        pkg.bam_header_from_schema(_schema({"bio.bam.reference_sequences": json.dumps([{"name": "z", "length": 0}])}))


def test_differential_fuzz_against_oracle(pkg, oracle):
    rng = random.Random(11)
    alpha = "abcXYZ019 _-:;,/\\\"é"

    def word():
        return "".join(rng.choice(alpha) for _ in range(rng.randrange(1, 9)))

    for _ in range(300):
        md = {}
        if rng.random() < 0.5:
            md["bio.bam.file_format_version"] = rng.choice(["1.0", "1.6", "2.11", "x", "1", "1.a"])
        for k in ("bio.bam.sort_order", "bio.bam.group_order", "bio.bam.subsort_order"):
            if rng.random() < 0.3:
                md[k] = word()
        if rng.random() < 0.7:
            md["bio.bam.reference_sequences"] = json.dumps(
                [dict({"name": word(), "length": rng.randrange(1, 1 << 31)},
                      **({"other_fields": {rng.choice(["AS", "M5", "UR", "SP", "QQ"]): word() for _ in range(rng.randrange(0, 3))}} if rng.random() < 0.4 else {}))
                 for _ in range(rng.randrange(0, 5))])
        if rng.random() < 0.5:
            md["bio.bam.read_groups"] = json.dumps(
                [dict({"id": word()}, **{k: word() for k in ("sample", "platform", "library", "description") if rng.random() < 0.5},
                      **({"other_fields": {rng.choice(["PU", "CN", "PM", "ZZ"]): word()}} if rng.random() < 0.4 else {}))
                 for _ in range(rng.randrange(0, 3))])
        if rng.random() < 0.5:
            md["bio.bam.program_info"] = json.dumps(
                [dict({"id": word()}, **{k: word() for k in ("name", "version", "command_line") if rng.random() < 0.5}) for _ in range(rng.randrange(0, 3))])
        if rng.random() < 0.4:
            md["bio.bam.comments"] = json.dumps([word() for _ in range(rng.randrange(0, 3))])
        sow = rng.choice([None, True, False])
        assert pkg.bam_header_from_schema(_schema(md), sow) == oracle.build_bam_header(md, sow), md


@pytest.mark.parametrize("name", ["multi_chrom.bam", "multi_chrom_large.bam", "bam_with_tags.bam", "10x_pbmc_tags.bam", "nanopore_custom_tags.bam", "no_coor_only.bam"])
def test_fixture_header_round_trips_through_the_metadata(pkg, oracle, name):
    path = os.path.join(G, name)
    if not os.path.exists(path):
        pytest.skip("fixture not present")
    orc = oracle.BamOracle(path)
    md = {k: v for k, v in (orc.schema.metadata or {}).items()}
    md = {k.decode(): v.decode() for k, v in md.items()}
    rebuilt = pkg.bam_header_from_schema(pa.schema([pa.field("name", pa.string())], metadata=md))
    assert rebuilt == oracle.build_bam_header(md)

    def norm(text):
        """record kind -> list of (sorted) field sets, fields the metadata does not carry dropped"""
        out = []
        for ln in text.rstrip("\n").split("\n"):
            if not ln:
                continue
            parts = ln.split("\t")
            if parts[0] == "@CO":
                out.append(("@CO", "\t".join(parts[1:])))
            else:
                out.append((parts[0], tuple(sorted(parts[1:]))))
        return out
    orig = norm(orc.hdr.text)
    got = norm(rebuilt)
    # every line kind and count survives; @SQ names / lengths, @RG / @PG ids and the fields the metadata structs hold are equal
    # (the rebuilt header orders its records @HD, @SQ, @RG, @PG, @CO as noodles writes them; a source file may interleave
    # them: the records of one kind keep their order)
    rank = {"@HD": 0, "@SQ": 1, "@RG": 2, "@PG": 3, "@CO": 4}
    orig = sorted(orig, key=lambda x: rank[x[0]])
    assert [k for k, _ in got if k != "@HD"] == [k for k, _ in orig if k != "@HD"]
    keep = {"@SQ": ("SN:", "LN:", "AH:", "AN:", "AS:", "DS:", "M5:", "SP:", "TP:", "UR:"),
            "@RG": ("ID:", "SM:", "PL:", "LB:", "DS:", "BC:", "CN:", "DT:", "FO:", "KS:", "PG:", "PI:", "PM:", "PU:"),
            "@PG": ("ID:", "PN:", "VN:", "CL:", "PP:", "DS:")}
    for (k, a), (_, b) in zip([x for x in orig if x[0] != "@HD"], [x for x in got if x[0] != "@HD"]):
        if k == "@CO":
            assert a == b
        else:
            assert tuple(f for f in a if f.startswith(keep[k])) == b, (k, a, b)
