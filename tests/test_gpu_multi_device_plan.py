"""One plan over several GPUs of a node (bioscan_scan_devices) and per-device byte-range residency (each GPU
receives only the compressed byte ranges its partitions cover).  The GPU box has one device, so a "two device" plan
names device 0 twice: the routing, the contiguous-run rule and the results are exercised; two PROVIDERS on the same file
stand in for two ranks and must hold disjoint compressed ranges."""
import os

import pytest

from test_gpu_bam_parity import _cmp_batches

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_open_leaves_nothing_resident(pkg):
    """The header is inflated from a private upload of the leading members; what becomes resident is decided by the
    partitions a provider executes."""
    prov = pkg.BamTableProvider(os.path.join(G, "multi_chrom_large.bam"))
    assert prov.resident_range(0) == (0, 0)


def test_plan_over_devices_routes_contiguous_runs_and_matches_oracle(pkg, oracle):
    path = os.path.join(G, "multi_chrom_large.bam")
    prov = pkg.BamTableProvider(path)
    orc = oracle.BamOracle(path)
    plan = prov.scan(target_partitions=8, device_ids=[0, 0, 0])
    parts, residual = orc.scan(target_partitions=8)
    n = plan.num_partitions()
    assert n == len(parts)
    weights = [plan.partition_estimated_bytes(p) for p in range(n)]
    runs = pkg.shard_partitions_in_order(weights, 3)
    assert [i for r in runs for i in r] == list(range(n))
    assert all(plan.partition_device(p) == 0 for p in range(n)) and plan.partition_device(n) == -1
    total = 0
    for p in range(n):
        got = list(plan.execute(p, 100))
        _, want = orc.execute_partition(parts[p].regions, None, residual, 100)
        _cmp_batches(got, want, ("devices", p))
        total += sum(b.num_rows for b in got)
    assert total == 4277


def test_two_providers_on_one_file_hold_disjoint_compressed_ranges(pkg, oracle):
    """Two ranks of `bench.py --mode indexed`: same file, the plan's partitions sharded in order; each rank makes only
    its own partitions resident.  Their resident ranges must not overlap beyond the shared header members, and the
    rows of rank 0 followed by rank 1 are the single-GPU rows."""
    path = os.path.join(G, "multi_chrom_large.bam")
    size = os.path.getsize(path)
    orc = oracle.BamOracle(path)
    parts, residual = orc.scan(filters=[("chrom", "in", ["chr1", "chr2", "chrX"])], target_partitions=6)
    ranges, rows = [], []
    for rank in range(2):
        prov = pkg.BamTableProvider(path)
        plan = prov.scan(filters=[("chrom", "in", ["chr1", "chr2", "chrX"])], target_partitions=6)
        n = plan.num_partitions()
        assert n == len(parts)
        mine = pkg.shard_partitions_in_order([plan.partition_estimated_bytes(p) for p in range(n)], 2)[rank]
        assert mine
        plan.make_resident(mine)
        lo, hi = prov.resident_range(0)
        assert 0 < hi - lo < size, "a rank uploads its own share, not the whole file"
        ranges.append((lo, hi))
        for p in mine:
            got = list(plan.execute(p, 64))
            _, want = orc.execute_partition(parts[p].regions, None, residual, 64)
            _cmp_batches(got, want, ("rank", rank, p))
            rows.append(sum(b.num_rows for b in got))
        assert prov.resident_range(0) == (lo, hi), "execute must not widen what make_resident uploaded"
    (lo0, hi0), (lo1, hi1) = ranges
    assert lo0 < lo1 and hi0 <= hi1   # contiguous runs in plan order (in a 34-member file the BAI's coarse bins make the spans wide)
    want_rows = 0
    for p in parts:
        bs = orc.execute_partition(p.regions, None, residual, 1 << 20)[1]
        want_rows += sum(b.num_rows for b in bs)
    assert sum(rows) == want_rows


def test_ranks_of_an_indexed_scan_hold_nearly_disjoint_ranges(pkg, tmp_path):
    """The same on a file large enough for the BAI chunks to be local (4096 members): four ranks, each makes only its
    run of the 32-partition plan resident; the ranges are in order, none is the whole file, neighbours overlap by a few
    members at most, and together the ranks return every record."""
    import json
    import subprocess
    exe = os.path.join(ROOT, "tools", "_build", "synth_bam")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools")])
    path = str(tmp_path / "s4096.bam")
    meta = json.loads(subprocess.check_output([exe, path, "4096", "11", "8"]).decode())
    size = os.path.getsize(path)
    world, spans, rows = 4, [], 0
    for rank in range(world):
        prov = pkg.BamTableProvider(path)
        plan = prov.scan(projection=[1, 2], target_partitions=8 * world)
        n = plan.num_partitions()
        mine = pkg.shard_partitions_in_order([plan.partition_estimated_bytes(p) for p in range(n)], world)[rank]
        plan.make_resident(mine)
        lo, hi = prov.resident_range(0)
        spans.append((lo, hi))
        for p in mine:
            rows += sum(b.num_rows for b in plan.execute(p, 8192))
        assert prov.resident_range(0) == (lo, hi)
    assert rows == meta["n_records"]
    member = size / 4096
    for (lo_a, hi_a), (lo_b, hi_b) in zip(spans, spans[1:]):
        assert lo_a < lo_b and hi_a <= hi_b
        assert hi_a - lo_b < 64 * member, (spans, "neighbouring ranks share more than a few members")
    # the last rank also scans the unplaced reads at the end of the file; no rank holds more than ~half of it
    assert all(hi - lo < 0.6 * size for lo, hi in spans), spans
