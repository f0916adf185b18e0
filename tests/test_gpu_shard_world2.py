"""The HIP compute path under more than one rank: two processes share the box's one GPU (a context each), every rank
runs EDT + A* of ITS block of the query list through the C ABI and packs its results with sc_gather_pack (device); the
messages travel between the processes with a gloo all_gather (RCCL refuses two ranks on one device; the RCCL transport
itself is covered at world size 1 by test_gpu_gather.py) and sc_gather_unpack on every rank must leave the whole batch, in
query order, equal to the oracle's single-process result -- the same pack / unpack kernels sc_allgather_paths runs around
its ncclAllGather."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, Q, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sea_current_amd as sc
        from sea_current_amd import shard, synth
        ctx = sc.Context(0)
        occ = synth.salt_grid(320, 256, 0.2, seed=31)
        d2 = ctx.edt(torch.from_numpy(occ).cuda())          # every rank recomputes the EDT locally
        torch.cuda.synchronize()
        s, g = synth.queries(d2.cpu().numpy() >= 1, Q, seed=7)
        Lmax = 1024

        q0, q1 = shard.rank_range(Q, world, rank)
        out = ctx.astar_batch(d2, torch.from_numpy(s[q0:q1]).cuda(), torch.from_numpy(g[q0:q1]).cuda(), Lmax=Lmax)
        cap = 16384
        msg = ctx.gather_pack(out, Q, world, rank, cap)
        torch.cuda.synchronize()
        words = msg.numel()
        all_msgs = torch.empty(world * words, dtype=torch.int32)
        dist.all_gather_into_tensor(all_msgs, msg.cpu())
        got = ctx.gather_unpack(all_msgs.cuda(), world, Q, Lmax, cap)
        torch.cuda.synchronize()
        ret[rank] = {k: v.cpu().numpy().copy() for k, v in got.items()}
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("Q", [64, 61])
def test_two_ranks_hip_compute(oracle, Q):
    import torch.multiprocessing as mp
    from sea_current_amd import synth
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), Q, ret), nprocs=world, join=True)
    occ = synth.salt_grid(320, 256, 0.2, seed=31)
    d2 = oracle.edt(occ)
    s, g = synth.queries(d2 >= 1, Q, seed=7)
    ref = oracle.astar_batch(d2, s, g, Lmax=1024)
    for r in range(world):
        o = ret[r]
        assert int(o["truncated"][0]) == 0
        for k in ("len", "cost", "status"):
            assert np.array_equal(o[k], ref[k]), (r, k)
        for q in range(Q):
            a, b = o["offsets"][q], o["offsets"][q + 1]
            assert np.array_equal(o["cells"][a:b], ref["path"][q, :ref["len"][q]] if ref["status"][q] == 0 else []), (r, q)
