"""What follows the planner on every path of the reference's examples (examples/zmq_test.cpp:66-93), for a batch of
paths at once, on the GPU through the C ABI:

    waypoints -> bezier_spline::from_path (sea_current.hpp:599-683) -> arclength (:767-896) -> gen_vel_prof<1> along the
    arclength (:1191-1265: TOPP-RA + sampling at dt) -> resample (:898-1005, nudge) -> curvature / angular velocity

`smooth_batch` is the batched form of that sequence; bench.py times it (leg `smoothing`) and tests/test_gpu_pipeline.py
checks it path by path against the same sequence on the CPU restatement.  The only work outside the library calls is bookkeeping between them
(the ragged profile samples packed back to back), done with torch on the device."""
import numpy as np


def waypoints_from_cells(path_cells, lens, W, n_wp=16, cell_m=0.05, jitter=0.2):
    """A* cell paths (numpy int32 [P, Lmax], len [P]) -> n_wp waypoints per path in metres (float32 [P, n_wp, 2]): the cells
    at equal fractions of the path, first and last cell included; callers leave out paths shorter than n_wp cells.
    Interior waypoints are moved by up to `jitter` cells (a hash of the cell index): the reference's waypoints are Halton
    samples, never collinear, and its tangent construction (sea_current.hpp:343-377) returns NaN on float32 triples that are
    collinear up to rounding (acos of a dot product just above 1) -- which grid paths are full of."""
    P = path_cells.shape[0]
    frac = np.linspace(0.0, 1.0, n_wp)
    idx = np.rint(frac[None, :] * (lens[:, None] - 1)).astype(np.int64)
    cells = np.take_along_axis(path_cells.astype(np.int64), idx, axis=1)
    h = cells.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)             # splitmix64 finaliser: no linear structure left
    h = (h ^ (h >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    h = (h ^ (h >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    h = h ^ (h >> np.uint64(31))
    jx = ((h & np.uint64(1023)).astype(np.float32) / np.float32(1023.0) - np.float32(0.5)) * np.float32(2.0 * jitter)
    jy = (((h >> np.uint64(10)) & np.uint64(1023)).astype(np.float32) / np.float32(1023.0) - np.float32(0.5)) * np.float32(2.0 * jitter)
    jx[:, [0, -1]] = 0.0
    jy[:, [0, -1]] = 0.0
    wp = np.stack([(cells % W).astype(np.float32) + jx, (cells // W).astype(np.float32) + jy], axis=-1) * np.float32(cell_m)
    return wp.reshape(P, n_wp, 2).astype(np.float32)


def smooth_batch(ctx, wp, vmax=1.0, amax=0.5, dt=0.02, N=100, nsub=100, max_len=None, nudge=True):
    """wp float32 GPU [P, n, 2] (n waypoints per path) -> dict with ctrl [P*(n-1),4,2], arclength [P], profile length [P],
    offsets [P+1], and per sample (packed back to back) pos, vel, pts [M,2], curvature, ang_vel.
    max_len: samples reserved per path (default: from the longest path's arclength at vmax / amax, with margin)."""
    import torch
    P, n, _ = wp.shape
    dev = wp.device
    npts = torch.full((P,), n, dtype=torch.int32, device=dev)
    ctrl = ctx.bezier_from_path(wp.contiguous(), npts)                       # [P, n-1, 4, 2]
    c2 = ctrl.reshape(-1, 4, 2)
    cum, seg_len = ctx.bezier_arclength(c2, nsub)                            # [S, nsub+1], [S]
    sl = seg_len.reshape(P, n - 1)
    AL = sl[:, 0].clone()
    for j in range(1, n - 1):                                                # float32, segment by segment, as the reference adds them (:896)
        AL = AL + sl[:, j]
    if max_len is None:
        al_max = float(AL.max())
        max_len = int((al_max / vmax + 2.0 * vmax / amax) / dt * 1.25) + 64
    z = torch.zeros((P, 1), dtype=torch.float64, device=dev)
    p1 = AL.double().reshape(P, 1)
    vlo = torch.full((P, 1), -vmax, dtype=torch.float64, device=dev)
    vhi = torch.full((P, 1), vmax, dtype=torch.float64, device=dev)
    alo = torch.full((P, 1), -amax, dtype=torch.float64, device=dev)
    ahi = torch.full((P, 1), amax, dtype=torch.float64, device=dev)
    res = ctx.toppra(z, p1, z, z, vlo, vhi, alo, ahi, N=N)
    smp = ctx.toppra_sample(z, p1, z, z, res["x"], res["t"], float(np.float32(dt)), max_len=max_len)
    length = smp["length"]
    # pack the ragged profiles back to back (what the reference does with one std::vector per path)
    keep = torch.arange(max_len, device=dev)[None, :] < length[:, None]
    pos = smp["pos"][:, 0, :][keep].contiguous()
    vel = smp["vel"][:, 0, :][keep].contiguous()
    off = torch.zeros(P + 1, dtype=torch.int32, device=dev)
    off[1:] = torch.cumsum(length, 0)
    seg_off = torch.arange(0, P * (n - 1) + 1, n - 1, dtype=torch.int32, device=dev)
    out = ctx.bezier_resample(c2, cum, AL, seg_off, pos, off, nudge=nudge, want_curvature=True)
    return dict(ctrl=c2, cum=cum, arclength=AL, toppra_status=res["status"], x=res["x"], t=res["t"], length=length, offsets=off,
                pos=pos, vel=vel, pts=out["pts"], curvature=out["curvature"], ang_vel=vel * out["curvature"],
                resample_status=out["status"], max_len=max_len)
