"""CPU: pin the FASTQ oracle against the reference's own tests
(bio-format-fastq/tests/parallel_read_test.rs, row_count_integration_test.rs, write_test.rs)."""
import os

import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def fo():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(G), "..", "oracle"))
    import fastq_oracle
    return fastq_oracle


def _rows(o, target, **kw):
    strat, parts = o.scan(target)
    rows = []
    for p in parts:
        _, bs = o.execute(strat, p, **kw)
        for b in bs:
            rows += list(zip(*[b.column(i).to_pylist() for i in range(b.num_columns)]))
    return strat, len(parts), rows


def test_bgzf_partition_counts(fo):
    o = fo.FastqOracle(os.path.join(G, "sample.fastq.bgz"))
    assert len(o.gzi) == 9 and o.gzi[0] == (13976 + 0 if False else o.gzi[0][0], o.gzi[0][1])
    base = None
    for target in range(1, 9):                                    # parallel_read_test.rs:22-45
        strat, nparts, rows = _rows(o, target)
        assert strat == "bgzf" and nparts == min(target, 10)
        assert len(rows) == 2000
        assert len({r[0] for r in rows}) == 2000                  # :48-105 no duplicate names
        base = base or sorted(rows)
        assert sorted(rows) == base                               # :190-232 same rows for any split


def test_uncompressed_splits(fo):
    o = fo.FastqOracle(os.path.join(G, "example.fastq"))
    for target in (1, 2, 3, 4, 8):                                # parallel_read_test.rs:237-414
        strat, nparts, rows = _rows(o, target)
        assert strat == ("sequential" if target == 1 else "byterange")
        assert len(rows) == 200


def test_limit_is_per_partition_upper_bound(fo):
    o = fo.FastqOracle(os.path.join(G, "sample.fastq.bgz"))
    strat, parts = o.scan(4)
    for p in parts:
        _, bs = o.execute(strat, p, limit=7)
        assert sum(b.num_rows for b in bs) <= 7                    # parallel_read_test.rs:134-155


def test_name_description_split(fo, tmp_path):
    p = tmp_path / "d.fastq"                                        # write_test.rs:217-286
    p.write_bytes(b"@seq_alpha description text here\nACGT\n+\nIIII\n@seq_beta\nAC\n+\nII\n")
    o = fo.FastqOracle(str(p))
    _, bs = o.execute("sequential", None)
    rows = bs[0].to_pylist()
    assert rows[0]["name"] == "seq_alpha" and rows[0]["description"] == "description text here"
    assert rows[1]["name"] == "seq_beta" and rows[1]["description"] is None   # NULL when empty (physical_exec.rs:430-434)


def test_gzi_bounds_kat(fo):
    # get_bgzf_partition_bounds: (0,0) is prepended, blocks split evenly, remainder to the first partitions
    gzi = [(100 * i, 1000 * i) for i in range(1, 10)]             # 10 blocks
    assert fo.bgzf_partition_bounds(gzi, 1) == [(0, None)]
    assert fo.bgzf_partition_bounds(gzi, 3) == [(0, 400), (4000, 700), (7000, None)]
    assert len(fo.bgzf_partition_bounds(gzi, 64)) == 10
    assert fo.bgzf_partition_bounds([], 4) == [(0, None)]


def test_c_fastq_oracle_matches_python_oracle(fo):
    """oracle/bioscan_oracle.c::oracle_fastq_scan_mem (bench.py's cpu_baseline for --format fastq) against the Python
    oracle on the reference's sample.fastq.bgz, for 1..5 threads (thread cuts use the reference's resync rule)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(G), "..", "oracle"))
    import c_oracle
    path = os.path.join(G, "sample.fastq.bgz")
    o = fo.FastqOracle(path)
    _, _, rows = _rows(o, 1)
    want = dict(n_rows=len(rows), name_bytes=sum(len(r[0].encode()) for r in rows),
                desc_bytes=sum(len(r[1].encode()) for r in rows if r[1] is not None), desc_null=sum(r[1] is None for r in rows),
                seq_bytes=sum(len(r[2]) for r in rows), qual_bytes=sum(len(r[3]) for r in rows),
                byte_sum=sum(sum(r[0].encode()) + 3 * (sum(r[1].encode()) if r[1] is not None else 0) + 5 * sum(r[2].encode()) + 7 * sum(r[3].encode()) for r in rows))
    data = open(path, "rb").read()
    for threads in (1, 2, 3, 5):
        got = c_oracle.fastq_scan(data, threads=threads)
        assert {k: got[k] for k in want} == want, (threads, got, want)
    assert got["n_rows"] == 2000
