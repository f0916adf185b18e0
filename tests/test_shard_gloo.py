"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the query sharding + result gather
(sea_current_amd/shard.py).  The compute step is injected (the CPU oracle) because the product
compute path is GPU-only; what is under test here is the partitioning, the two gather wire forms and
rank-count independence of the results."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, Q, compact, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sea_current_amd import shard, synth
        from oracle import oracle
        occ = synth.salt_grid(96, 80, 0.15, seed=21)
        d2 = oracle.edt(occ)          # every rank recomputes the EDT locally (no broadcast)
        s, g = synth.queries(d2 >= 1, Q, seed=5)
        Lmax = 400

        def plan_fn(s_loc, g_loc):
            r = oracle.astar_batch(d2, s_loc.numpy(), g_loc.numpy(), Lmax=Lmax)
            return {k: torch.from_numpy(r[k]) for k in ("path", "len", "cost", "status")}

        out = shard.plan_sharded(plan_fn, torch.from_numpy(s), torch.from_numpy(g), world, rank, dist, Lmax=Lmax, compact=compact)
        ret[rank] = {k: v.numpy().copy() for k, v in out.items()}
    finally:
        dist.destroy_process_group()


def _reference(Q):
    from sea_current_amd import synth
    from oracle import oracle
    occ = synth.salt_grid(96, 80, 0.15, seed=21)
    d2 = oracle.edt(occ)
    s, g = synth.queries(d2 >= 1, Q, seed=5)
    return oracle.astar_batch(d2, s, g, Lmax=400)


def _run(world, Q, compact):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), Q, compact, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


@pytest.mark.parametrize("world,Q", [(2, 24), (3, 24)])
def test_fixed_stride_allgather(world, Q, oracle):
    ref = _reference(Q)
    outs = _run(world, Q, compact=False)
    for o in outs:  # every rank holds all results, in query order, identical to a single-rank run
        for k in ("len", "cost", "status"):
            assert np.array_equal(o[k], ref[k]), k
        for q in range(Q):
            assert np.array_equal(o["path"][q, :ref["len"][q]], ref["path"][q, :ref["len"][q]])


@pytest.mark.parametrize("world,Q", [(2, 24), (2, 23), (3, 10)])
def test_compact_allgather_ragged(world, Q, oracle):
    ref = _reference(Q)
    outs = _run(world, Q, compact=True)
    for o in outs:
        for k in ("len", "cost", "status"):
            assert np.array_equal(o[k], ref[k]), k
        assert o["offsets"][-1] == ref["len"][ref["status"] == 0].sum()
        for q in range(Q):
            a, b = o["offsets"][q], o["offsets"][q + 1]
            if ref["status"][q] == 0:
                assert np.array_equal(o["cells"][a:b], ref["path"][q, :ref["len"][q]])
            else:
                assert a == b


def test_rank_range_partitions():
    from sea_current_amd import shard
    for Q in (0, 1, 7, 1024, 65536 + 3):
        for world in (1, 2, 3, 8):
            r = [shard.rank_range(Q, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == Q
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_queries_are_rank_count_independent():
    from sea_current_amd import synth
    trav = synth.salt_grid(64, 64, 0.1, seed=2) == 0
    s, g = synth.queries(trav, 40)
    s2, g2 = synth.queries(trav, 15, first=25)
    assert np.array_equal(s[25:], s2) and np.array_equal(g[25:], g2)
