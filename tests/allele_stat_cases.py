"""Known answers of vcf_an / vcf_ac / vcf_af, transcribed from the reference's unit tests
(bio-format-vcf/src/udfs.rs:1165-1553), shared by the oracle (CPU) and the product (GPU) tests."""
import pyarrow as pa

GT_T = pa.list_(pa.field("item", pa.utf8(), True))

# test_parse_gt_alleles (:1165-1190)
PARSE_KATS = [("0/1", [0, 1]), ("1|0", [1, 0]), (".", None), ("./.", None), (".|.", None), ("0", [0]), ("0/1/2", [0, 1, 2]), ("./1", [None, 1])]
# test_count_alt_alleles (:1555-1562)
ALT_KATS = [("A", 1), ("A|T", 2), ("A|T|C", 3), ("", 0), (".", 0), ("  A|T  ", 2)]

# (name, GT rows, ALT (scalar broadcast or None), AN, AC, AF)
STAT_KATS = [
    # make_test_ctx rows (test_vcf_an / _ac / _af, :1193-1270)
    ("test_data", [["0/1", "1/1", "0/0"], ["./.", "0/1", "1/1"]], None, [6, 4], [[3], [3]], [[0.5], [0.75]]),
    # test_allele_stats_multiallelic (:1273-1377): 1- and 2-argument forms agree
    ("multiallelic", [["0/2", "1/2", "0/1"]], None, [6], [[2, 2]], [[1 / 3, 1 / 3]]),
    ("multiallelic, ALT", [["0/2", "1/2", "0/1"]], "A|T", [6], [[2, 2]], [[1 / 3, 1 / 3]]),
    # test_allele_stats_all_missing (:1380-1476)
    ("all missing", [["./.", "./."]], None, [0], [[]], [[]]),
    ("all missing, ALT", [["./.", "./."]], "A|T", [0], [[0, 0]], [[None, None]]),
    # test_ac_af_two_arg_unobserved_alt (:1480-1553, issue #103)
    ("unobserved alt", [["0/1", "0/1", "1/1"]], None, [4 + 2], [[4]], [[4 / 6]]),
    ("unobserved alt, ALT", [["0/1", "0/1", "1/1"]], "A|T", [6], [[4, 0]], [[4 / 6, 0.0]]),
]


def arrays(rows, alt):
    gt = pa.array(rows, type=GT_T)
    return gt, (None if alt is None else pa.array([alt] * len(rows), type=pa.utf8()))
