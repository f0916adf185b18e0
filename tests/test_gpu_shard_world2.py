"""The HIP compute path under more than one rank: two processes share the box's one GPU (a context each), every rank
runs EDT + A* of ITS block of the query list through the C ABI, the blocks are gathered (gloo here: RCCL refuses two
ranks on one device; the RCCL transport itself is covered at world size 1 by test_gpu_gather.py) and every rank must
hold the whole batch, in query order, equal to the oracle's single-process result."""
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

        def plan_fn(s_loc, g_loc):
            out = ctx.astar_batch(d2, s_loc.cuda(), g_loc.cuda(), Lmax=Lmax)
            torch.cuda.synchronize()
            return {k: v.cpu() for k, v in out.items()}

        out = shard.plan_sharded(plan_fn, torch.from_numpy(s), torch.from_numpy(g), world, rank, dist, Lmax=Lmax, compact=True)
        ret[rank] = {k: v.numpy().copy() for k, v in out.items()}
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
        for k in ("len", "cost", "status"):
            assert np.array_equal(o[k], ref[k]), (r, k)
        for q in range(Q):
            a, b = o["offsets"][q], o["offsets"][q + 1]
            assert np.array_equal(o["cells"][a:b], ref["path"][q, :ref["len"][q]] if ref["status"][q] == 0 else []), (r, q)
