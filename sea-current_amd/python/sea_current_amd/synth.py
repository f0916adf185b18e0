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


def block_rects(W, H, coverage=0.20, seed=SEED_GRID, smin=4, smax=64):
    """The rectangle list behind block_grid: int32 [R,4] = (x0, y0, x1, y1), x1/y1 exclusive, clipped to the grid."""
    occ = np.zeros((H, W), dtype=np.uint8)
    target = int(coverage * W * H)
    rects = []
    covered = 0
    while covered < target:
        i = len(rects)
        r = _u01(seed ^ 0xB10C, np.arange(4 * i, 4 * i + 4, dtype=np.uint64))
        w = smin + int(r[0] * (smax - smin + 1)); h = smin + int(r[1] * (smax - smin + 1))
        x0 = int(r[2] * W); y0 = int(r[3] * H)
        occ[y0:y0 + h, x0:x0 + w] = 1
        rects.append((x0, y0, min(x0 + w, W), min(y0 + h, H)))
        if len(rects) % 64 == 0 or covered == 0:
            covered = int(occ.sum())
    return np.array(rects, dtype=np.int32)


def raster_rects(rects, W, H, base=None, free_border=True):
    """occ uint8 [H,W]: `base` (or an empty grid) with the rectangles painted as occupied, borders free."""
    occ = np.zeros((H, W), dtype=np.uint8) if base is None else np.array(base, dtype=np.uint8)
    for x0, y0, x1, y1 in np.asarray(rects).tolist():
        occ[max(y0, 0):max(y1, 0), max(x0, 0):max(x1, 0)] = 1
    return _free_border(occ) if free_border else occ


def block_grid(W, H, coverage=0.20, seed=SEED_GRID, smin=4, smax=64):
    """Random axis-aligned rectangles (side smin..smax) until >= coverage, borders free."""
    return raster_rects(block_rects(W, H, coverage, seed, smin, smax), W, H)


def move_rects(rects, frame, W, H, K=32, step=2, seed=SEED_GRID):
    """Frame `frame` of the dynamic-obstacle stream: K of the rectangles (chosen by hash of the frame
    number) move by at most `step` cells in x and y, staying inside the grid.  Returns the new list; frames are applied
    one after the other starting from block_rects()."""
    out = np.array(rects, dtype=np.int32)
    R = out.shape[0]
    h = splitmix64(np.uint64(seed ^ 0xD1A) ^ splitmix64(np.uint64(frame) * np.uint64(3 * K) + np.arange(3 * K, dtype=np.uint64)))
    for k in range(min(K, R)):
        i = int(h[3 * k] % np.uint64(R))
        dx = int(h[3 * k + 1] % np.uint64(2 * step + 1)) - step
        dy = int(h[3 * k + 2] % np.uint64(2 * step + 1)) - step
        x0, y0, x1, y1 = out[i]
        dx = max(-x0, min(dx, W - x1)); dy = max(-y0, min(dy, H - y1))
        out[i] = (x0 + dx, y0 + dy, x1 + dx, y1 + dy)
    return out


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


def polygon_world(n_poly=14, half=5.0, seed=SEED_GRID):
    """Convex polygons (3 .. 6 corners, radius 0.3 .. 1.0) scattered over [-0.8 half, 0.8 half]^2 -> (lines float32 [E,4]
    (x0, y0, x1, y1 per edge), obs_off int32 [n_poly+1]): the obstacle form of the reference's planning_space
    (sea_current.hpp:193-284), for the FMT* leg of the bench and its tests."""
    rng = np.random.default_rng(seed)
    lines, off = [], [0]
    for _ in range(n_poly):
        c = rng.uniform(-0.8 * half, 0.8 * half, 2)
        r = rng.uniform(0.3, 1.0)
        ang = np.sort(rng.uniform(0, 2 * np.pi, int(rng.integers(3, 7))))
        q = [(c[0] + r * np.cos(a), c[1] + r * np.sin(a)) for a in ang]
        for i in range(len(q)):
            lines.append(tuple(q[i]) + tuple(q[(i + 1) % len(q)]))
        off.append(len(lines))
    return np.array(lines, np.float32).reshape(-1, 4), np.array(off, np.int32)


def _halton(base, n, skip=20):
    out = np.empty(n, np.float64)
    for k in range(n):
        f, r, i = 1.0, 0.0, k + 1 + skip
        while i > 0:
            f /= base
            r += f * (i % base)
            i //= base
        out[k] = r
    return out


def free_samples(n, half, lines, obs_off, seed=0):
    """n Halton points (bases 2, 3) of [-half, half]^2 outside every polygon (even-odd ray casting) -> float32 [n,2].  A
    generator of inputs for the bench: the reference's own sample_free (:1294-1313) is restated in oracle/ and in the header."""
    pts = []
    k = 0
    while len(pts) < n:
        m = 2 * (n - len(pts)) + 16
        hx = _halton(2, m + k)[k:]
        hy = _halton(3, m + k)[k:]
        k += m
        P = np.stack([(2 * hx - 1) * half, (2 * hy - 1) * half], axis=1)
        inside = np.zeros(P.shape[0], bool)
        for o in range(obs_off.shape[0] - 1):
            cnt = np.zeros(P.shape[0], np.int32)
            for (x0, y0, x1, y1) in lines[obs_off[o]:obs_off[o + 1]].astype(np.float64):
                cross = ((y0 > P[:, 1]) != (y1 > P[:, 1])) & (P[:, 0] < (x1 - x0) * (P[:, 1] - y0) / (y1 - y0 + 1e-300) + x0)
                cnt += cross
            inside |= (cnt & 1) == 1
        pts.extend(P[~inside].tolist())
    return np.array(pts[:n], np.float32)
