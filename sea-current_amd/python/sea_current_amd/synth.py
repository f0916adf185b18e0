"""Bit-reproducible synthetic workloads.

Everything is a pure function of (seed, index) through splitmix64, so query i /
cell i / plan i is the same on any rank count and on CPU or GPU hosts.
"""
import numpy as np

SEED_GRID = 0xC0FFEE
SEED_QUERY = 0xBEEF
SEED_TOPPRA = 0x70BBA

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _u01(seed, idx):
    h = splitmix64(np.uint64(seed) ^ splitmix64(np.asarray(idx, dtype=np.uint64)))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _free_border(occ):
    occ[0, :] = 0; occ[-1, :] = 0; occ[:, 0] = 0; occ[:, -1] = 0
    return occ


def salt_grid(W, H, p=0.05, seed=SEED_GRID):
    """i.i.d. obstacles: occ = (u < p), borders free.  uint8 [H, W]."""
    u = _u01(seed, np.arange(W * H, dtype=np.uint64)).reshape(H, W)
    return _free_border((u < p).astype(np.uint8))


def block_grid(W, H, coverage=0.20, seed=SEED_GRID, smin=4, smax=64):
    """Random axis-aligned rectangles (side smin..smax) until >= coverage, borders free."""
    occ = np.zeros((H, W), dtype=np.uint8)
    target = int(coverage * W * H)
    i = 0
    covered = 0
    while covered < target:
        r = _u01(seed ^ 0xB10C, np.arange(4 * i, 4 * i + 4, dtype=np.uint64))
        w = smin + int(r[0] * (smax - smin + 1)); h = smin + int(r[1] * (smax - smin + 1))
        x0 = int(r[2] * W); y0 = int(r[3] * H)
        occ[y0:y0 + h, x0:x0 + w] = 1
        i += 1
        if i % 64 == 0 or covered == 0:
            covered = int(occ.sum())
    return _free_border(occ)


def largest_component(trav):
    """Boolean mask of the largest 4-connected component of `trav` (no-corner-cutting
    8-connected reachability == 4-connected reachability)."""
    from scipy import ndimage
    lab, n = ndimage.label(trav)
    if n == 0:
        return np.zeros_like(trav, dtype=bool)
    cnt = np.bincount(lab.ravel())
    cnt[0] = 0
    return lab == int(np.argmax(cnt))


def queries(trav, Q, seed=SEED_QUERY, first=0):
    """Start/goal linear cell indices for queries first..first+Q-1, drawn uniformly from
    the largest component of the traversable mask `trav` [H, W].  int32 arrays."""
    cells = np.flatnonzero(largest_component(np.asarray(trav, dtype=bool)).ravel())
    n = cells.shape[0]
    if n == 0:
        raise ValueError("no traversable cell")
    i = np.arange(first, first + Q, dtype=np.uint64)
    hs = splitmix64(np.uint64(seed) ^ splitmix64(np.uint64(2) * i))
    hg = splitmix64(np.uint64(seed) ^ splitmix64(np.uint64(2) * i + np.uint64(1)))
    return (cells[(hs % np.uint64(n)).astype(np.int64)].astype(np.int32),
            cells[(hg % np.uint64(n)).astype(np.int64)].astype(np.int32))


def toppra_plans(P, dof=6, seed=SEED_TOPPRA, first=0):
    """Hermite endpoints/tangents + limits for plans first..first+P-1:
    q0,q1 ~ U[-pi,pi]^dof, tangents ~ U[-1,1]^dof, v_lim = +-[2,2,2,3,3,3],
    a_lim = +-[5,5,5,8,8,8] (pattern repeated/truncated for other dof)."""
    idx = (np.arange(first, first + P, dtype=np.uint64)[:, None, None] * np.uint64(4 * dof)
           + np.arange(4, dtype=np.uint64)[None, :, None] * np.uint64(dof)
           + np.arange(dof, dtype=np.uint64)[None, None, :])
    u = _u01(seed, idx)
    p0 = (2 * u[:, 0] - 1) * np.pi
    p1 = (2 * u[:, 1] - 1) * np.pi
    v0 = 2 * u[:, 2] - 1
    v1 = 2 * u[:, 3] - 1
    vl = np.resize(np.array([2, 2, 2, 3, 3, 3.0]), dof)
    al = np.resize(np.array([5, 5, 5, 8, 8, 8.0]), dof)
    return dict(p0=p0, p1=p1, v0=v0, v1=v1, vlim=np.tile(vl, (P, 1)), alim=np.tile(al, (P, 1)))
