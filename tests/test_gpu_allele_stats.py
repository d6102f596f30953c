"""vcf_an / vcf_ac / vcf_af on the device (bioscan_udf_vcf_allele_stats) against the reference's unit tests
(bio-format-vcf/src/udfs.rs:1165-1562, tests/allele_stat_cases.py) and, on random genotype columns, against the oracle."""
import random

import pyarrow as pa
import pytest

import allele_stat_cases as K
from conftest import load_oracle  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def V():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import vcf_oracle
    return vcf_oracle


def test_reference_kats(pkg):
    for name, rows, alt, an, ac, af in K.STAT_KATS:
        gt, alt_arr = K.arrays(rows, alt)
        got_an = pkg.vcf_an(gt)
        assert got_an.type == pa.int32() and got_an.to_pylist() == an, name
        got_ac = pkg.vcf_ac(gt, alt_arr)
        assert got_ac.type == pa.list_(pa.field("item", pa.int32(), True)) and got_ac.to_pylist() == ac, name
        got_af = pkg.vcf_af(gt, alt_arr)
        assert got_af.type == pa.list_(pa.field("item", pa.float64(), True)), name
        for g, w in zip(got_af.to_pylist(), af):
            assert len(g) == len(w) and all((a is None and b is None) or (a is not None and b is not None and abs(a - b) < 0.001) for a, b in zip(g, w)), (name, g, w)


def test_parse_rules_through_the_kernels(pkg, V):
    """Every rule of parse_gt_alleles: one genotype per row, so AN / AC show what the device made of it."""
    gts = ["0/1", "1|0", ".", "./.", ".|.", "0", "0/1/2", "./1", " 1/2 ", "1 / 2", "+1/+2", "-1/2", "1/x", "", "/", "1//2", "01/002",
           "18446744073709551615/1", "18446744073709551616/1", "./.|.", "3", "\t2|2\n", "1/.", "0|0|0|7", "1/2/"]
    gt = pa.array([[g] for g in gts] + [None, [], [None], [None, "1/1"]], type=K.GT_T)
    for fn in ("vcf_an",):
        assert getattr(pkg, fn)(gt).to_pylist() == getattr(V, fn)(gt).to_pylist()
    # the two huge indices would make the reference allocate a vector of that length: both sides refuse / are skipped here
    small = pa.array([[g] for g in gts if not g.startswith("1844")] + [None, [], [None], [None, "1/1"]], type=K.GT_T)
    assert pkg.vcf_ac(small).to_pylist() == V.vcf_ac(small).to_pylist()
    assert pkg.vcf_af(small).to_pylist() == V.vcf_af(small).to_pylist()
    with pytest.raises(pkg.BioscanError, match="out of range"):
        pkg.vcf_ac(pa.array([["18446744073709551615/1"]], type=K.GT_T))


def test_random_columns_match_the_oracle(pkg, V):
    rng = random.Random(5)

    def gt():
        r = rng.random()
        if r < 0.08:
            return None
        if r < 0.16:
            return rng.choice([".", "./.", ".|."])
        ploidy = rng.choice([1, 2, 2, 2, 3])
        sep = rng.choice("/|")
        return sep.join("." if rng.random() < 0.1 else str(rng.choice([0, 0, 0, 1, 1, 2, 3, 5])) for _ in range(ploidy))
    for n_samples in (1, 3, 64, 65, 1000):
        rows, alts = [], []
        for _ in range(40):
            rows.append(None if rng.random() < 0.05 else [gt() for _ in range(rng.choice([0, n_samples]) if rng.random() < 0.1 else n_samples)])
            alts.append(rng.choice([None, "", ".", "A", "A|T", "A|T|C|G|AA|<DEL>", " G|C "]))
        g = pa.array(rows, type=K.GT_T)
        a = pa.array(alts, type=pa.utf8())
        assert pkg.vcf_an(g).equals(V.vcf_an(g)), n_samples
        assert pkg.vcf_ac(g).equals(V.vcf_ac(g)), n_samples
        assert pkg.vcf_ac(g, a).equals(V.vcf_ac(g, a)), n_samples
        assert pkg.vcf_af(g).equals(V.vcf_af(g)), n_samples           # (the division is the same IEEE operation on both sides)
        assert pkg.vcf_af(g, a).equals(V.vcf_af(g, a)), n_samples
        # sliced inputs carry offsets
        assert pkg.vcf_ac(g.slice(7, 20), a.slice(7, 20)).equals(V.vcf_ac(g.slice(7, 20), a.slice(7, 20)))


def test_argument_errors(pkg):
    with pytest.raises(pkg.BioscanError, match="expects List<Utf8> input"):
        pkg.vcf_an(pa.array([[1, 2]], type=pa.list_(pa.int32())))
    with pytest.raises(pkg.BioscanError, match="2nd argument must be Utf8"):
        pkg.vcf_ac(pa.array([["0/1"]], type=K.GT_T), pa.array([1]))


def test_genotypes_column_of_a_scan(pkg, V, tmp_path):
    """The UDFs on what the VCF scan produces: genotypes.GT of a multi-sample file, with the scan's own alt column (ALT alleles
    joined by '|', the form count_alt_alleles expects)."""
    lines = ["##fileformat=VCFv4.3", "##contig=<ID=chr1,length=100000>", '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
             "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2\tS3\tS4"]
    rows = [("A", "G", ["0/1", "1/1", "0/0", "./."]), ("C", "T,G", ["0/2", "1|2", "./1", "0/0"]), ("G", "A,C,T", ["0/1", "0/1", "1/1", "."]),
            ("T", ".", ["0/0", "0/0", "./.", "0|0"]), ("A", "AT", [".", "./.", ".|.", "./."])]
    for k, (ref, alt, gts) in enumerate(rows):
        lines.append("\t".join(["chr1", str(100 + k), ".", ref, alt, "50", "PASS", ".", "GT"] + gts))
    path = tmp_path / "ms.vcf"
    path.write_text("\n".join(lines) + "\n")
    prov = pkg.VcfTableProvider(str(path))
    plan = prov.scan()
    t = pa.Table.from_batches([b for p in range(plan.num_partitions()) for b in plan.execute(p, 8192)])
    gt = t["genotypes"].combine_chunks().field("GT")
    alt = t["alt"].combine_chunks()
    assert alt.to_pylist() == ["G", "T|G", "A|C|T", "", "AT"]
    assert pkg.vcf_an(gt).to_pylist() == [6, 7, 6, 6, 0] == V.vcf_an(gt).to_pylist()
    assert pkg.vcf_ac(gt, alt).to_pylist() == [[3], [2, 2], [4, 0, 0], [], [0]] == V.vcf_ac(gt, alt).to_pylist()
    assert pkg.vcf_af(gt, alt).equals(V.vcf_af(gt, alt))
    assert pkg.vcf_af(gt, alt).to_pylist()[4] == [None]
