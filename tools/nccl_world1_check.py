"""Scratch: exercise the RCCL code path of shard.allgather_paths / allgather_paths_compact with a world of one rank
(the only RCCL configuration a one-GPU box can run): API usage, dtypes, contiguity.  Launch with
python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 tools/nccl_world1_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sea-current_amd", "python")); sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.distributed as dist
import sea_current_amd as sc
from sea_current_amd import synth, shard
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
ctx = sc.Context(0)
occ = synth.salt_grid(256, 256, 0.2)
d2 = ctx.edt(torch.from_numpy(occ).cuda()); torch.cuda.synchronize()
s, g = synth.queries(d2.cpu().numpy() >= 1, 64)
out = ctx.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), Lmax=1024)
torch.cuda.synchronize()
gathered = shard.allgather_paths(out, shard.alloc_gather(out, 1), dist)
torch.cuda.synchronize()
for k in ("len", "cost", "status", "path"):
    assert torch.equal(gathered[k], out[k]), k
comp = shard.allgather_paths_compact(out, dist, 1)
assert torch.equal(comp["len"], out["len"]) and int(comp["offsets"][-1]) == int(out["len"][out["status"] == 0].sum())
tt = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
dist.barrier()
dist.destroy_process_group()
print("nccl world-1 check ok")
