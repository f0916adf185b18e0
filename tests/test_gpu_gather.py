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
        assert ctx.allgather_last_bytes() == 4 * int(sc.lib().sc_gather_msg_words(200, 1, total + 7))
        # a message too small for the paths: flagged, and what did fit is still right
        small = ctx.allgather_paths(out, 200, cap_cells=total // 2)
        torch.cuda.synchronize()
        assert int(small["truncated"][0]) == 3      # bit 0: the message was too small; bit 1: so is the compact array (world * cap_cells)
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


def _virtual_ranks_case(ctx, oracle, torch, out_all, ref, Q, world, Lmax, cap_cells, cells_capacity=None):
    """Play `world` ranks on one GPU: sc_gather_pack per rank block -> messages back to back in rank order (device copies)
    -> sc_gather_unpack; every output against the oracle's single-process result."""
    import sea_current_amd as sc
    import ctypes as C
    words = int(sc.lib().sc_gather_msg_words(Q, world, cap_cells))
    qmax = -(-Q // world)
    assert words >= 2 + 3 * qmax + cap_cells and words % 2 == 0
    msgs = torch.full((world * words,), -99, dtype=torch.int32, device="cuda")
    trunc_ranks = []
    for r in range(world):
        a, b = C.c_int(), C.c_int()
        sc.lib().sc_rank_range(Q, world, r, C.byref(a), C.byref(b))
        q0, q1 = a.value, b.value
        blk = {k: out_all[k][q0:q1].contiguous() for k in ("path", "len", "cost", "status")}
        ctx.gather_pack(blk, Q, world, r, cap_cells, msg=msgs[r * words:(r + 1) * words], Lmax=Lmax)
        eff = np.where(ref["status"][q0:q1] == 0, ref["len"][q0:q1], 0).sum()
        if eff > cap_cells:
            trunc_ranks.append(r)
    got = ctx.gather_unpack(msgs, world, Q, Lmax, cap_cells, want_path=True, cells_capacity=cells_capacity)
    torch.cuda.synchronize()
    for k in ("len", "cost", "status"):
        assert np.array_equal(got[k].cpu().numpy(), ref[k][:Q]), (world, k)
    eff = np.where(ref["status"][:Q] == 0, ref["len"][:Q], 0).astype(np.int64)
    off = got["offsets"].cpu().numpy()
    assert np.array_equal(off, np.concatenate([[0], np.cumsum(eff)])), world
    cells = got["cells"].cpu().numpy()
    path = got["path"].cpu().numpy()
    cc = cells.shape[0]
    flag = int(got["truncated"][0])
    assert (flag & 1) == (1 if trunc_ranks else 0), (world, flag, trunc_ranks)
    assert (flag & 2) == (2 if off[Q] > cc else 0), (world, flag)
    for r in range(world):
        a, b = C.c_int(), C.c_int()
        sc.lib().sc_rank_range(Q, world, r, C.byref(a), C.byref(b))
        sent = 0     # cells of this rank's message in front of the query
        for q in range(a.value, b.value):
            L = int(eff[q])
            want = ref["path"][q, :L].copy()
            n_ok = max(0, min(L, cap_cells - sent))      # what fitted into the rank's message
            want[n_ok:] = -1
            assert np.array_equal(path[q, :L], want), (world, r, q)
            lo, hi = off[q], min(off[q] + L, cc)
            if hi > lo:
                assert np.array_equal(cells[lo:hi], want[:hi - lo]), (world, r, q)
            sent += L
    return flag


def test_gather_virtual_ranks_against_oracle(oracle):
    """The rank lookup of gather_scan_kernel and gather_unpack_kernel's per-rank offsets with r > 0: 2, 3 and 8 messages
    built from blocks of one oracle-checked batch (Q % world != 0, ranks without a query, a rank over cap_cells, a compact
    array that is too small)."""
    import torch
    import sea_current_amd as sc
    from sea_current_amd import synth
    ctx = sc.Context(0)
    try:
        occ = synth.block_grid(320, 256, 0.2, seed=5)
        d2h = oracle.edt(occ)
        Q = 203
        s, g = synth.queries(d2h >= 1, Q, seed=11)
        s[17] = -1                                               # bad endpoint: no cells travel
        g[40] = int(np.flatnonzero(occ.ravel())[3])
        Lmax = 1024
        d2 = ctx.edt(torch.from_numpy(occ).cuda())
        out = ctx.astar_batch(d2, torch.from_numpy(s).cuda(), torch.from_numpy(g).cuda(), Lmax=Lmax)
        torch.cuda.synchronize()
        ref = oracle.astar_batch(d2h, s, g, Lmax=Lmax, nthreads=4)
        for k in ("len", "cost", "status"):
            assert np.array_equal(out[k].cpu().numpy(), ref[k]), k
        eff = np.where(ref["status"] == 0, ref["len"], 0)
        for world in (1, 2, 3, 8):
            # roomy messages
            blocks = [eff[q0:q1].sum() for q0, q1 in (_rr(Q, world, r) for r in range(world))]
            assert _virtual_ranks_case(ctx, oracle, torch, out, ref, Q, world, Lmax, int(max(blocks)) + 3) == 0
            # the largest block does not fit (odd capacity: the message stride is rounded to even), the others do
            if world > 1:
                second = sorted(blocks)[-2]
                cap = int((max(blocks) + second) // 2) | 1
                assert second <= cap < max(blocks)
                assert _virtual_ranks_case(ctx, oracle, torch, out, ref, Q, world, Lmax, cap) == 1
            # compact array smaller than the cells of all paths
            assert _virtual_ranks_case(ctx, oracle, torch, out, ref, Q, world, Lmax, int(max(blocks)) + 3,
                                       cells_capacity=int(eff.sum()) - 5) == 2
        # fewer queries than ranks: ranks 5..7 own nothing and still send a header
        sub = {k: out[k][:5].contiguous() for k in out}
        ref5 = {k: ref[k][:5] for k in ref}
        assert _virtual_ranks_case(ctx, oracle, torch, sub, ref5, 5, 8, Lmax, int(eff[:5].max()) + 1) == 0
    finally:
        ctx.close()


def _rr(Q, world, r):
    base, rem = divmod(Q, world)
    q0 = r * base + min(r, rem)
    return q0, q0 + base + (1 if r < rem else 0)


def test_gather_pack_rejects_wrong_block(oracle):
    import torch
    import sea_current_amd as sc
    ctx = sc.Context(0)
    try:
        out = dict(path=torch.zeros((4, 8), dtype=torch.int32, device="cuda"), len=torch.zeros(4, dtype=torch.int32, device="cuda"),
                   cost=torch.zeros(4, dtype=torch.int32, device="cuda"), status=torch.zeros(4, dtype=torch.int32, device="cuda"))
        with pytest.raises(sc.SeaCurrentError):
            ctx.gather_pack(out, 10, 2, 0, 16)          # rank 0 of 2 owns 5 of 10 queries, not 4
        assert sc.lib().sc_gather_msg_words(0, 2, 16) == 0
    finally:
        ctx.close()
