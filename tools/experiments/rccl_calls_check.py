"""Dev check: the torch.distributed calls bench.py makes at N > 1 (RCCL backend, device_id, barrier, f64 MAX / SUM
all_reduce, broadcast_object_list, destroy), run with one rank so it fits a one-GPU box.
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/experiments/rccl_calls_check.py"""
import os

import torch
import torch.distributed as dist

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
dist.barrier()
torch.cuda.synchronize()
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
c = torch.tensor([3.0, 4.0, 5.0], dtype=torch.float64, device="cuda")
dist.all_reduce(c, op=dist.ReduceOp.SUM)
assert t.item() == 1.25 and c.tolist() == [3.0, 4.0, 5.0]
# the start-up broadcast of the shared input (config 5): path, generator metadata, member count
box = ["/dev/shm/x.bam", {"n_blocks": 5, "n_records": 7}, 655360]
dist.broadcast_object_list(box, src=0, device=torch.device("cuda", local_rank))
assert box[2] == 655360 and box[1]["n_records"] == 7
dist.barrier()
dist.destroy_process_group()
print("rccl calls ok")
