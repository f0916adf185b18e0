"""Query sharding across ranks (one process per GPU) and the gather of result paths.

Independent start-goal queries (and TOPP-RA plans) shard with no data-path collective: query i goes
to rank i*R//Q in contiguous blocks, so gathered outputs are already in query order.
The grid is replicated and every rank recomputes the EDT locally (tens of microseconds; cheaper than
broadcasting 4 B/cell over xGMI).  The only exchange is the final all-gather of the result paths,
through torch.distributed -- backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests.

Two wire forms:
  allgather_paths          fixed stride [Q_local, Lmax] int32 (one collective; parity format)
  allgather_paths_compact  lengths first, then only sum(len) cells per rank padded to the max
                           payload over ranks (what a 4096^2 / 64k-query job wants: Lmax >> mean len)
"""
import numpy as np


def rank_range(Q, world, rank):
    """Contiguous block [q0, q1) of rank `rank`; blocks differ by at most one query."""
    base, rem = divmod(Q, world)
    q0 = rank * base + min(rank, rem)
    return q0, q0 + base + (1 if rank < rem else 0)


def alloc_gather(out, world):
    """Receive buffers for allgather_paths: same keys as `out`, leading dim * world."""
    import torch
    return {k: torch.empty((v.shape[0] * world,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
            for k, v in out.items()}


def allgather_paths(out, gathered, dist, group=None):
    """Fixed-stride all-gather of path/len/cost/status.  Every rank must hold the same Q_local.  Two collectives per
    call: the three per-query int32 arrays travel as one [3, Q_local] block, the paths as the other."""
    import torch
    Q = out["len"].shape[0]
    world = gathered["len"].shape[0] // max(Q, 1)
    meta = torch.stack([out["len"], out["cost"], out["status"]])                     # [3, Q]
    meta_all = torch.empty((world, 3, Q), dtype=meta.dtype, device=meta.device)
    dist.all_gather_into_tensor(meta_all.view(-1), meta.contiguous().view(-1), group=group)
    for j, k in enumerate(("len", "cost", "status")):
        gathered[k].view(world, Q).copy_(meta_all[:, j, :])
    dist.all_gather_into_tensor(gathered["path"].view(-1), out["path"].contiguous().view(-1), group=group)
    return gathered


def pack_paths(path, length):
    """[Q, Lmax] + len -> (flat cells of all paths back to back, offsets [Q+1])."""
    import torch
    Q, Lmax = path.shape
    ln = length.clamp(min=0, max=Lmax).to(torch.int64)
    offs = torch.zeros(Q + 1, dtype=torch.int64, device=path.device)
    offs[1:] = torch.cumsum(ln, 0)
    mask = torch.arange(Lmax, device=path.device)[None, :] < ln[:, None]
    return path[mask], offs


def allgather_paths_compact(out, dist, world, group=None):
    """Variable-length gather: returns (len_all [Q], cost_all, status_all, flat cells, offsets [Q+1])
    for ALL queries in query order.  Works for unequal Q_local (pads to the max over ranks)."""
    import torch
    dev = out["len"].device
    qloc = torch.tensor([out["len"].shape[0]], dtype=torch.int64, device=dev)
    qs = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(qs, qloc, group=group)
    qmax = int(qs.max())
    flat, offs = pack_paths(out["path"], torch.where(out["status"] == 0, out["len"], torch.zeros_like(out["len"])))
    nloc = torch.tensor([flat.shape[0]], dtype=torch.int64, device=dev)
    ns = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(ns, nloc, group=group)
    nmax = max(int(ns.max()), 1)

    def padded(v, n, fill=0):
        p = torch.full((n,), fill, dtype=v.dtype, device=dev)
        p[:v.shape[0]] = v
        return p

    meta = torch.stack([padded(out[k], qmax) for k in ("len", "cost", "status")])       # [3, qmax]
    meta_all = torch.empty((world,) + tuple(meta.shape), dtype=meta.dtype, device=dev)
    dist.all_gather_into_tensor(meta_all.view(-1), meta.contiguous().view(-1), group=group)
    cells_all = torch.empty((world, nmax), dtype=flat.dtype, device=dev)
    dist.all_gather_into_tensor(cells_all.view(-1), padded(flat, nmax, -1), group=group)
    qs_l, ns_l = qs.tolist(), ns.tolist()
    ln = torch.cat([meta_all[r, 0, :qs_l[r]] for r in range(world)])
    cost = torch.cat([meta_all[r, 1, :qs_l[r]] for r in range(world)])
    status = torch.cat([meta_all[r, 2, :qs_l[r]] for r in range(world)])
    cells = torch.cat([cells_all[r, :ns_l[r]] for r in range(world)])
    eff = torch.where(status == 0, ln, torch.zeros_like(ln)).to(torch.int64)
    offsets = torch.zeros(eff.shape[0] + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(eff, 0)
    return dict(len=ln, cost=cost, status=status, cells=cells, offsets=offsets)


def plan_sharded(plan_fn, start, goal, world, rank, dist=None, Lmax=4096, compact=False, group=None):
    """Run `plan_fn(start_local, goal_local) -> dict(path,len,cost,status)` on this rank's block of the
    global query list and gather every rank's results.  `plan_fn` is the HIP path in production
    (Context.astar_batch); the CPU tests inject the oracle to exercise the sharding logic under gloo."""
    Q = start.shape[0]
    q0, q1 = rank_range(Q, world, rank)
    out = plan_fn(start[q0:q1], goal[q0:q1])
    if world == 1 or dist is None:
        return out
    if compact or Q % world != 0:
        return allgather_paths_compact(out, dist, world, group=group)
    return allgather_paths(out, alloc_gather(out, world), dist, group=group)
