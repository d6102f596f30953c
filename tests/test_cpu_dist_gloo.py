"""CPU, world_size 2 over gloo: the N>1 bench path = independent shards + MAX-over-ranks time +
SUM of units, with NO collective on the data path."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg, load_oracle
    pkg, oracle = load_pkg(), load_oracle()
    path = os.path.join(ROOT, "tests", "golden", "multi_chrom_large.bam")
    b = oracle.BamOracle(path)
    parts, res = b.scan(target_partitions=6)
    mine = pkg.shard_partitions_in_order([p.total_estimated_bytes for p in parts], world)[rank]
    rows = 0
    names = []
    for p in mine:                                  # each rank scans only its own partitions
        _, bs = b.execute_partition(parts[p].regions, [0], res, 8192)
        for x in bs:
            rows += x.num_rows
            names += x.column(0).to_pylist()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)   # stand-in for the per-rank elapsed time
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(rows)], dtype=torch.float64)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    q.put((rank, mine, rows, float(t.item()), float(c.item())))
    dist.destroy_process_group()


def test_two_rank_sharded_scan():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    (r0, m0, n0, t0, c0), (r1, m1, n1, t1, c1) = out
    assert m0 + m1 == list(range(len(m0) + len(m1))) and m0 and m1   # contiguous runs, rank order = partition order
    assert n0 + n1 == 4277 and c0 == c1 == 4277.0
    assert t0 == t1 == 2.0


def _vcf_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from conftest import load_pkg
    import vcf_oracle
    pkg = load_pkg()
    o = vcf_oracle.VcfOracle(os.path.join(ROOT, "tests", "golden", "multi_chrom_large.vcf.gz"))
    plan = o.scan(target_partitions=5)                       # TBI chunk ranges -> 5 balanced partitions
    mine = pkg.shard_partitions_in_order([a.total_estimated_bytes for a in plan["assignments"]], world)[rank]
    starts = []
    for p in mine:
        _, bs = o.execute(plan, p, 8192)
        for x in bs:
            starts += list(zip(x.column("chrom").to_pylist(), x.column("start").to_pylist()))
    c = torch.tensor([float(len(starts))], dtype=torch.float64)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    q.put((rank, mine, starts, float(c.item())))
    dist.destroy_process_group()


def test_two_rank_sharded_vcf_scan():
    """VCF: TBI partitions shard across ranks with no exchange; rank-order concatenation = single-rank row order."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_vcf_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
    (r0, m0, s0, c0), (r1, m1, s1, c1) = out
    assert m0 + m1 == list(range(len(m0) + len(m1))) and m0 and m1
    assert c0 == c1 == 10000.0
    rows = s0 + s1
    assert len(set(rows)) == 10000 and rows == sorted(rows, key=lambda t: (t[0], t[1]))


def _fastq_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from conftest import load_pkg
    import fastq_oracle
    pkg = load_pkg()
    o = fastq_oracle.FastqOracle(os.path.join(ROOT, "tests", "golden", "sample.fastq.bgz"))
    strat, parts = o.scan(6)                                  # GZI block ranges -> 6 partitions
    mine = pkg.shard_partitions_in_order([1] * len(parts), world)[rank]
    names = []
    for p in mine:
        _, bs = o.execute(strat, parts[p])
        for x in bs:
            names += x.column("name").to_pylist()
    c = torch.tensor([float(len(names))], dtype=torch.float64)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    q.put((rank, strat, mine, names, float(c.item())))
    dist.destroy_process_group()


def test_two_rank_sharded_fastq_scan():
    """FASTQ: GZI partitions shard across ranks with no exchange; every read is owned by exactly one rank and the
    rank-order concatenation is the file order (fastq/tests/parallel_read_test.rs: 2000 reads, no duplicates)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_fastq_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
    (r0, st0, m0, n0, c0), (r1, st1, m1, n1, c1) = out
    assert st0 == st1 == "bgzf"
    assert m0 + m1 == list(range(len(m0) + len(m1))) and m0 and m1
    assert c0 == c1 == 2000.0
    assert len(set(n0 + n1)) == 2000
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fastq_oracle
    o = fastq_oracle.FastqOracle(os.path.join(ROOT, "tests", "golden", "sample.fastq.bgz"))
    strat, parts = o.scan(1)
    whole = [n for b in o.execute(strat, parts[0])[1] for n in b.column("name").to_pylist()]
    assert n0 + n1 == whole
