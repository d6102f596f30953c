"""Partition -> GPU sharding: contiguous runs of partitions in reference order, byte
balanced, so that concatenating rank 0..N-1 reproduces the single-GPU partition order.  Same rule
as `partition_byte_ranges_in_order` (bio-format-core/src/range_planning.rs:147-195) applied to
the plan's per-partition byte estimates.  No collective is involved: ranks never exchange data."""
from typing import List, Sequence


def shard_partitions_in_order(weights: Sequence[int], world_size: int) -> List[List[int]]:
    """weights[i] = estimated bytes of partition i.  Returns, per rank, the list of partition
    indices it executes (a contiguous run; ranks beyond the partition count get [])."""
    n = len(weights)
    world = max(1, world_size)
    if n == 0:
        return [[] for _ in range(world)]
    count = min(world, n)
    total = sum(weights)
    runs, cur, assigned = [], [], 0
    for idx, w in enumerate(weights):
        remaining_ranges = n - idx
        remaining_parts = count - len(runs)
        share_complete = count > 1 and cur and assigned * count >= total * (len(runs) + 1)
        must_close = remaining_ranges < remaining_parts
        if remaining_parts > 1 and (share_complete or must_close):
            runs.append(cur)
            cur = []
        cur.append(idx)
        assigned += w
    runs.append(cur)
    while len(runs) < world:
        runs.append([])
    return runs
