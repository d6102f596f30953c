"""world_size 2 over gloo with the PRODUCT on the data path (the CPU twin, tests/test_cpu_dist_gloo.py, can only run the
oracle): two processes share the one GPU of the box, each opens the same BAM through the C ABI, takes its run of the
BAI plan (`shard_partitions_in_order`), uploads only what that run inflates (`make_resident`) and scans it; the bench's
own collectives (barrier, MAX of times, SUM of rows) run over gloo.  No collective touches the data path."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_pkg, load_oracle
    pkg, oracle = load_pkg(), load_oracle()
    path = os.path.join(ROOT, "tests", "golden", "multi_chrom_large.bam")
    prov = pkg.BamTableProvider(path)
    plan = prov.scan(projection=[0, 1, 2], target_partitions=6)
    orc = oracle.BamOracle(path)
    parts, res = orc.scan(target_partitions=6)
    n = plan.num_partitions()
    mine = pkg.shard_partitions_in_order([plan.partition_estimated_bytes(p) for p in range(n)], world)[rank]
    plan.make_resident(mine)
    lo, hi = prov.resident_range(0)
    rows, ok = 0, True
    for p in mine:
        got = list(plan.execute(p, 500))
        _, want = orc.execute_partition(parts[p].regions, [0, 1, 2], res, 500)
        ok = ok and len(got) == len(want) and all(g.equals(w) for g, w in zip(got, want))
        rows += sum(b.num_rows for b in got)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([float(rows)], dtype=torch.float64)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    q.put((rank, mine, rows, ok, float(t.item()), float(c.item()), lo, hi, os.path.getsize(path)))
    dist.destroy_process_group()


def test_two_ranks_scan_their_runs_through_the_c_abi():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(60)
    (r0, m0, n0, ok0, t0, c0, lo0, hi0, size), (r1, m1, n1, ok1, t1, c1, lo1, hi1, _) = out
    assert ok0 and ok1                                                 # every partition oracle-equal on its rank
    assert m0 + m1 == list(range(len(m0) + len(m1))) and m0 and m1     # contiguous runs, rank order = partition order
    assert n0 + n1 == 4277 and c0 == c1 == 4277.0
    assert t0 == t1 == 2.0
    assert hi0 - lo0 < size and hi1 - lo1 < size and lo0 < lo1        # each rank holds its own byte range, in order
