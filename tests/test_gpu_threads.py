"""Concurrent partitions of ONE plan, one OS thread per partition, through the C ABI.

The reference polls the partitions of a plan concurrently: every `execute(partition)` owns its reader and runs its
blocking decode on whichever worker polls it (bio-format-core/src/sync_stream.rs:19-29,
bio-format-bam/src/physical_exec.rs:878-881).  `include/bioscan.h` promises the same: distinct streams may be
driven from distinct threads.  ctypes releases the GIL around every foreign call, so the threads below really run
`bioscan_execute` / `bioscan_next` of different partitions at the same time.  Each partition's batches must equal the
oracle's, exactly as in the single-threaded parity tests."""
import os
import sys
import threading

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
ROUNDS = 6  # every round restarts all partitions at once: several chances for any interleaving to show


def _run_threads(plan, n_parts, bs):
    """Executes partitions 0..n_parts-1 of `plan`, one thread each, all released by one barrier."""
    out = [None] * n_parts
    errs = []
    gate = threading.Barrier(n_parts)

    def work(p):
        try:
            gate.wait()
            out[p] = list(plan.execute(p, bs))
        except BaseException as e:  # noqa: BLE001 -- reported by the caller
            errs.append((p, repr(e)))

    ts = [threading.Thread(target=work, args=(p,)) for p in range(n_parts)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    return out


def _same(got, want, ctx):
    assert len(got) == len(want), (ctx, len(got), len(want))
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.num_rows == w.num_rows, (ctx, i, g.num_rows, w.num_rows)
        for n in w.schema.names:
            assert g.column(n).equals(w.column(n)), (ctx, i, n)


@pytest.mark.parametrize("fname,tags,target", [
    ("multi_chrom_large.bam", None, 8),
    ("multi_chrom.bam", None, 4),
    ("10x_pbmc_tags.bam", ["CB", "CR", "UB", "NH", "AS", "RG"], 4),
])
def test_bam_partitions_from_threads(pkg, oracle, fname, tags, target):
    path = os.path.join(G, fname)
    prov = pkg.BamTableProvider(path, None, True, tags)
    orc = oracle.BamOracle(path, zero_based=True, tag_fields=tags)
    plan = prov.scan(target_partitions=target)
    parts, residual = orc.scan(target_partitions=target)
    n = plan.num_partitions()
    assert n == len(parts) and n >= 2
    want = [orc.execute_partition(parts[p].regions, None, residual, 64)[1] for p in range(n)]
    for r in range(ROUNDS):
        got = _run_threads(plan, n, 64)
        for p in range(n):
            _same(got[p], want[p], (fname, "round", r, "partition", p, plan.partition_desc(p)))


def test_bam_two_plans_of_one_provider_from_threads(pkg, oracle):
    """Different plans (different projections and filters) of one provider, executed at the same time."""
    path = os.path.join(G, "multi_chrom_large.bam")
    prov = pkg.BamTableProvider(path)
    orc = oracle.BamOracle(path)
    plan_a = prov.scan(target_partitions=4)
    plan_b = prov.scan(projection=[1, 2, 9], filters=[("chrom", "=", "chr1")], target_partitions=3)
    parts_a, res_a = orc.scan(target_partitions=4)
    parts_b, res_b = orc.scan(filters=[("chrom", "=", "chr1")], target_partitions=3)
    want_a = [orc.execute_partition(p.regions, None, res_a, 100)[1] for p in parts_a]
    want_b = [orc.execute_partition(p.regions, [1, 2, 9], res_b, 100)[1] for p in parts_b]
    for r in range(3):
        res = {}

        def run(tag, plan, n):
            res[tag] = _run_threads(plan, n, 100)

        ta = threading.Thread(target=run, args=("a", plan_a, plan_a.num_partitions()))
        tb = threading.Thread(target=run, args=("b", plan_b, plan_b.num_partitions()))
        ta.start(); tb.start(); ta.join(); tb.join()
        for p, w in enumerate(want_a):
            _same(res["a"][p], w, ("plan a", r, p))
        for p, w in enumerate(want_b):
            _same(res["b"][p], w, ("plan b", r, p))


def test_fastq_partitions_from_threads(pkg):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fastq_oracle as fo
    path = os.path.join(G, "sample.fastq.bgz")
    orc = fo.FastqOracle(path)
    strat, parts = orc.scan(8)
    prov = pkg.FastqTableProvider(path)
    plan = prov.scan(target_partitions=8)
    n = plan.num_partitions()
    assert n == len(parts) and n >= 4
    want = [orc.execute(strat, part, batch_size=300)[1] for part in parts]
    for r in range(ROUNDS):
        got = _run_threads(plan, n, 300)
        for p in range(n):
            _same(got[p], want[p], ("fastq", r, p))
        assert sum(b.num_rows for bs in got for b in bs) == 2000


def test_vcf_partitions_from_threads(pkg):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import vcf_oracle as vo
    path = os.path.join(G, "multi_chrom_large.vcf.gz")
    o = vo.VcfOracle(path)
    g = pkg.VcfTableProvider(path)
    oplan = o.scan(target_partitions=4)
    gplan = g.scan(target_partitions=4)
    n = gplan.num_partitions()
    assert n == o.num_partitions(oplan) and n >= 2
    want = [o.execute(oplan, p, 500)[1] for p in range(n)]
    for r in range(ROUNDS):
        got = _run_threads(gplan, n, 500)
        total = 0
        for p in range(n):
            _same(got[p], want[p], ("vcf", r, p))
            total += sum(b.num_rows for b in got[p])
        assert total == 10000  # vcf/tests/indexed_read_test.rs: 5000 + 5000
