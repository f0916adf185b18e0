"""GPU: the HIP planning path through the C-ABI gather (sc_allgather_paths over RCCL) at world size 1, against the oracle:
pack -> ncclAllGather -> unpack must hand back exactly the paths sc_astar_batch produced, in CSR and fixed-stride form."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_allgather_paths_world1_matches_oracle(oracle):
    import torch
    import sea_current_amd as sc
    from sea_current_amd import synth
    ctx = sc.Context(0)
    try:
        ctx.comm_init(sc.Context.comm_unique_id(), 1, 0)
        occ = synth.salt_grid(256, 256, 0.2, seed=3)
        d2h = oracle.edt(occ)
        s, g = synth.queries(d2h >= 1, 200, seed=3)
        # a blocked goal and an out-of-range start among them: paths that must not travel
        s[5] = -1
        g[9] = int(np.flatnonzero(occ.ravel())[0])
        Lmax = 1024
        d2 = ctx.edt(torch.from_numpy(occ).cuda())
        out = ctx.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), Lmax=Lmax)
        ref = oracle.astar_batch(d2h, s, g, Lmax=Lmax, nthreads=4)
        total = int(ref["len"][ref["status"] == 0].sum())
        got = ctx.allgather_paths(out, 200, cap_cells=total + 7, want_path=True)
        torch.cuda.synchronize()
        assert int(got["truncated"][0]) == 0
        for k in ("len", "cost", "status"):
            assert np.array_equal(got[k].cpu().numpy(), ref[k]), k
        off = got["offsets"].cpu().numpy()
        eff = np.where(ref["status"] == 0, ref["len"], 0)
        assert np.array_equal(off, np.concatenate([[0], np.cumsum(eff)]))
        cells = got["cells"].cpu().numpy()
        path = got["path"].cpu().numpy()
        for q in range(200):
            if ref["status"][q] == 0:
                L = ref["len"][q]
                assert np.array_equal(cells[off[q]:off[q] + L], ref["path"][q, :L]), q
                assert np.array_equal(path[q, :L], ref["path"][q, :L]), q
        assert ctx.allgather_last_bytes() == 4 * (2 + 3 * 200 + total + 7)
        # a message too small for the paths: flagged, and what did fit is still right
        small = ctx.allgather_paths(out, 200, cap_cells=total // 2)
        torch.cuda.synchronize()
        assert int(small["truncated"][0]) == 1
        c2 = small["cells"].cpu().numpy()
        assert np.array_equal(c2[:total // 2], cells[:total // 2])
    finally:
        ctx.close()


def test_rank_range_partitions(oracle):
    import ctypes as C
    import sea_current_amd as sc
    from sea_current_amd import shard
    l = sc.lib()
    for Q, world in ((1024, 8), (1000, 3), (5, 8), (65536, 8)):
        prev = 0
        for r in range(world):
            a, b = C.c_int(), C.c_int()
            l.sc_rank_range(Q, world, r, C.byref(a), C.byref(b))
            assert (a.value, b.value) == shard.rank_range(Q, world, r) and a.value == prev
            prev = b.value
        assert prev == Q
