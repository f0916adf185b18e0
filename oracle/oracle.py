"""ctypes wrapper over oracle/libsc_oracle.so -- CPU ORACLE (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (sea-current_amd/) never does.
See oracle/sc_oracle.h for the semantics and the parity status of each function.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EDT_INF = 2**31 - 1
G_INF = 0xFFFFFFFF
OK, NO_PATH, BAD_ENDPOINT, PATH_TRUNCATED = 0, 1, 2, 3


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libsc_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("SC_ORACLE_LIB") or os.path.join(_HERE, "libsc_oracle.so")   # SC_ORACLE_LIB: the sanitizer build (make asan)
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.sco_astar.restype = C.c_int
        _LIB.sco_toppra.restype = C.c_int
        _LIB.sco_toppra_sample.restype = C.c_int
        _LIB.sco_bezier_arclength.restype = C.c_double
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def edt(occ, exact=True):
    occ = np.ascontiguousarray(occ, dtype=np.uint8)
    H, W = occ.shape
    d2 = np.empty((H, W), dtype=np.int32)
    fn = lib().sco_edt_exact if exact else lib().sco_edt_brute
    fn(_p(occ, C.c_uint8), C.c_int(W), C.c_int(H), _p(d2, C.c_int32))
    return d2


def edt_nearest(occ, d2=None):
    """Index of the nearest occupied cell per cell (ties: smallest index; -1: empty grid).  d2=None: brute force."""
    occ = np.ascontiguousarray(occ, dtype=np.uint8)
    H, W = occ.shape
    out = np.empty((H, W), dtype=np.int32)
    if d2 is None:
        lib().sco_edt_nearest_brute(_p(occ, C.c_uint8), C.c_int(W), C.c_int(H), _p(out, C.c_int32))
    else:
        d2 = np.ascontiguousarray(d2, dtype=np.int32)
        lib().sco_edt_nearest(_p(occ, C.c_uint8), _p(d2, C.c_int32), C.c_int(W), C.c_int(H), _p(out, C.c_int32))
    return out


def moves(d2, r2=0):
    d2 = np.ascontiguousarray(d2, dtype=np.int32)
    H, W = d2.shape
    m = np.empty((H, W), dtype=np.uint8)
    lib().sco_moves(_p(d2, C.c_int32), C.c_int(W), C.c_int(H), C.c_int32(r2), _p(m, C.c_uint8))
    return m


def astar(d2, start, goal, r2=0, Lmax=None, want_g=False):
    d2 = np.ascontiguousarray(d2, dtype=np.int32)
    H, W = d2.shape
    if Lmax is None:
        Lmax = W * H
    path = np.full(Lmax, -1, dtype=np.int32)
    g = np.empty((H, W), dtype=np.uint32) if want_g else None
    ln, cost, ex = C.c_int32(0), C.c_int32(-1), C.c_int64(0)
    st = lib().sco_astar(_p(d2, C.c_int32), C.c_int(W), C.c_int(H), C.c_int32(r2), C.c_int32(int(start)),
                         C.c_int32(int(goal)), C.c_int(Lmax), _p(path, C.c_int32), C.byref(ln), C.byref(cost),
                         _p(g, C.c_uint32), C.byref(ex))
    out = dict(status=st, len=ln.value, cost=cost.value, expanded=ex.value,
               path=path[:min(ln.value, Lmax)].copy() if st in (OK,) else path[:0].copy())
    if want_g:
        out["g"] = g
    return out


def astar_batch(d2, start, goal, r2=0, Lmax=4096, nthreads=1):
    d2 = np.ascontiguousarray(d2, dtype=np.int32)
    H, W = d2.shape
    start = np.ascontiguousarray(start, dtype=np.int32)
    goal = np.ascontiguousarray(goal, dtype=np.int32)
    Q = start.shape[0]
    path = np.full((Q, Lmax), -1, dtype=np.int32)
    ln = np.zeros(Q, dtype=np.int32)
    cost = np.zeros(Q, dtype=np.int32)
    status = np.zeros(Q, dtype=np.int32)
    ex = np.zeros(Q, dtype=np.int64)
    lib().sco_astar_batch(_p(d2, C.c_int32), C.c_int(W), C.c_int(H), C.c_int32(r2), _p(start, C.c_int32),
                          _p(goal, C.c_int32), C.c_int(Q), C.c_int(Lmax), _p(path, C.c_int32), _p(ln, C.c_int32),
                          _p(cost, C.c_int32), _p(status, C.c_int32), _p(ex, C.c_int64), C.c_int(nthreads))
    return dict(path=path, len=ln, cost=cost, status=status, expanded=ex)


def toppra(p0, p1, v0, v1, vlim_lo, vlim_hi, alim_lo, alim_hi, N=100, sd_start=0.0, sd_end=0.0):
    """One plan.  vlim_*: [N+1, dof] (or [dof], broadcast); alim_*: [dof]."""
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    p0, p1, v0, v1, alim_lo, alim_hi = map(f, (p0, p1, v0, v1, alim_lo, alim_hi))
    dof = p0.shape[0]
    vlim_lo = f(np.broadcast_to(f(vlim_lo), (N + 1, dof)))
    vlim_hi = f(np.broadcast_to(f(vlim_hi), (N + 1, dof)))
    K = np.zeros((N + 1, 2)); x = np.zeros(N + 1); u = np.zeros(N); t = np.zeros(N + 1)
    d = C.c_double
    st = lib().sco_toppra(C.c_int(dof), C.c_int(N), _p(p0, d), _p(p1, d), _p(v0, d), _p(v1, d), _p(vlim_lo, d),
                          _p(vlim_hi, d), _p(alim_lo, d), _p(alim_hi, d), d(sd_start), d(sd_end),
                          _p(K, d), _p(x, d), _p(u, d), _p(t, d))
    return dict(status=st, K=K, x=x, u=u, t=t)


def toppra_sample(p0, p1, v0, v1, x, t, dt, max_len=None):
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    p0, p1, v0, v1, x, t = map(f, (p0, p1, v0, v1, x, t))
    dof = p0.shape[0]
    N = x.shape[0] - 1
    if max_len is None:
        max_len = int(np.ceil(t[-1] / dt)) + 1
    pos = np.zeros((dof, max_len), dtype=np.float32)
    vel = np.zeros_like(pos); acc = np.zeros_like(pos)
    times = np.zeros(max_len)
    d = C.c_double
    n = lib().sco_toppra_sample(C.c_int(dof), C.c_int(N), _p(p0, d), _p(p1, d), _p(v0, d), _p(v1, d), _p(x, d),
                                _p(t, d), d(dt), C.c_int(max_len), _p(pos, C.c_float), _p(vel, C.c_float),
                                _p(acc, C.c_float), _p(times, d))
    m = min(n, max_len)
    return dict(length=n, pos=pos[:, :m], vel=vel[:, :m], acc=acc[:, :m], time=times[:m])


def bezier_from_path(path, start_angle=float("nan"), lines=None):
    """path [n,2] float32 -> ctrl [n-1,4,2] float32 (sea_current.hpp:599-683)."""
    path = np.ascontiguousarray(path, dtype=np.float32)
    n = path.shape[0]
    lines = np.zeros((0, 4), np.float32) if lines is None else np.ascontiguousarray(lines, dtype=np.float32)
    ctrl = np.zeros((n - 1, 4, 2), dtype=np.float32)
    lib().sco_bezier_from_path(_p(path, C.c_float), C.c_int(n), C.c_float(start_angle), _p(lines, C.c_float),
                               C.c_int(lines.shape[0]), _p(ctrl, C.c_float))
    return ctrl


def bezier_shrink_tangent(T, Wp, k, lines):
    """T, Wp [M,2] float32, lines [E,4] -> shrunk tangents [M,2] (sea_current.hpp:575-596)."""
    T = np.ascontiguousarray(T, dtype=np.float32)
    Wp = np.ascontiguousarray(Wp, dtype=np.float32)
    lines = np.ascontiguousarray(lines, dtype=np.float32).reshape(-1, 4)
    out = np.zeros_like(T)
    lib().sco_bezier_shrink_tangent(_p(T, C.c_float), _p(Wp, C.c_float), C.c_int(T.shape[0]), C.c_float(k), _p(lines, C.c_float),
                                    C.c_int(lines.shape[0]), _p(out, C.c_float))
    return out


def bezier_eval(ctrl, seg, t, order=0):
    ctrl = np.ascontiguousarray(ctrl, dtype=np.float32)
    seg = np.ascontiguousarray(seg, dtype=np.int32)
    t = np.ascontiguousarray(t, dtype=np.float64)
    out = np.zeros((t.shape[0], 2))
    lib().sco_bezier_eval(_p(ctrl, C.c_float), _p(seg, C.c_int32), _p(t, C.c_double), C.c_int(t.shape[0]), C.c_int(order),
                          _p(out, C.c_double))
    return out


def bezier_arclength(ctrl, nsub=100):
    ctrl = np.ascontiguousarray(ctrl, dtype=np.float32)
    nseg = ctrl.shape[0]
    cum = np.zeros((nseg, nsub + 1))
    total = lib().sco_bezier_arclength(_p(ctrl, C.c_float), C.c_int(nseg), C.c_int(nsub), _p(cum, C.c_double))
    return total, cum


def bezier_resample(ctrl, cum, arclength, profile_pos, nudge=True):
    """resample (sea_current.hpp:898-1005) -> dict(status, pos (nudged), pts [n,2], t [n], seg [n], curvature [n])."""
    ctrl = np.ascontiguousarray(ctrl, dtype=np.float32)
    cum = np.ascontiguousarray(cum, dtype=np.float32)
    nseg, m = cum.shape
    pp = np.array(profile_pos, dtype=np.float32)
    n = pp.shape[0]
    pts = np.zeros((n, 2), np.float32)
    tpar = np.zeros(n, np.float32)
    seg = np.zeros(n, np.int32)
    curv = np.zeros(n, np.float32)
    f = lib().sco_bezier_resample
    f.restype = C.c_int
    st = f(_p(ctrl, C.c_float), C.c_int(nseg), C.c_int(m - 1), _p(cum, C.c_float), C.c_float(float(arclength)), _p(pp, C.c_float),
           C.c_int(n), C.c_int(1 if nudge else 0), _p(pts, C.c_float), _p(tpar, C.c_float), _p(seg, C.c_int32), _p(curv, C.c_float))
    return dict(status=st, pos=pp, pts=pts, t=tpar, seg=seg, curvature=curv)


def bezier_curve(ctrl, seg, t):
    """general-degree curve; ctrl [nseg, deg+1, 2] -> points [m, 2] (float64)."""
    ctrl = np.ascontiguousarray(ctrl, dtype=np.float32)
    seg = np.ascontiguousarray(seg, dtype=np.int32)
    t = np.ascontiguousarray(t, dtype=np.float64)
    out = np.zeros((t.shape[0], 2))
    lib().sco_bezier_curve(_p(ctrl, C.c_float), C.c_int(ctrl.shape[1] - 1), _p(seg, C.c_int32), _p(t, C.c_double), C.c_int(t.shape[0]),
                           _p(out, C.c_double))
    return out


def chebfit(x, y, degree):
    """free chebfit (sea_current.hpp:1109-1138) -> (coef float64 [degree], xmin, xmax)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    coef = np.zeros(degree)
    xmin, xmax = C.c_double(), C.c_double()
    lib().sco_chebfit(_p(x, C.c_float), _p(y, C.c_float), C.c_int(x.shape[0]), C.c_int(degree), _p(coef, C.c_double), C.byref(xmin), C.byref(xmax))
    return coef, xmin.value, xmax.value


def chebeval(x, coef, xmin, xmax):
    x = np.ascontiguousarray(x, dtype=np.float32)
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    y = np.zeros(x.shape[0])
    lib().sco_chebeval(_p(x, C.c_float), C.c_int(x.shape[0]), C.c_int(coef.shape[0]), _p(coef, C.c_double), C.c_double(xmin), C.c_double(xmax),
                       _p(y, C.c_double))
    return y


def halton(b, n, state=(0, 0)):
    """next n base-b Halton numbers from state (f, i) -> (float32 [n], new state)."""
    f, i = C.c_int(state[0]), C.c_int(state[1])
    out = np.zeros(n, np.float32)
    lib().sco_halton(C.c_int(b), C.c_int(n), C.byref(f), C.byref(i), _p(out, C.c_float))
    return out, (f.value, i.value)


def sample_free(n, rect, lines, obs_off, hstate=(0, 0, 0, 0)):
    """-> (pts float32 [n,2], new hstate); rect = (x_min, x_max, y_min, y_max)."""
    rect = np.asarray(rect, np.float32)
    lines = np.ascontiguousarray(lines, np.float32).reshape(-1, 4)
    obs_off = np.ascontiguousarray(obs_off, np.int32)
    hs = np.array(hstate, np.int32)
    pts = np.zeros((n, 2), np.float32)
    f = lib().sco_sample_free
    f.restype = C.c_int
    st = f(C.c_int(n), _p(rect, C.c_float), _p(lines, C.c_float), _p(obs_off, C.c_int32), C.c_int(obs_off.shape[0] - 1), _p(hs, C.c_int32),
           _p(pts, C.c_float))
    if st:
        raise RuntimeError("sample_free: no free point")
    return pts, tuple(int(v) for v in hs)


def fmt_star(samples, start, goal, rn, lines, Lmax=256):
    samples = np.ascontiguousarray(samples, np.float32)
    lines = np.ascontiguousarray(lines, np.float32).reshape(-1, 4)
    path = np.zeros((Lmax, 2), np.float32)
    ln, cost = C.c_int32(0), C.c_float(0)
    f = lib().sco_fmt_star
    f.restype = C.c_int
    st = f(_p(samples, C.c_float), C.c_int(samples.shape[0]), C.c_float(start[0]), C.c_float(start[1]), C.c_float(goal[0]),
           C.c_float(goal[1]), C.c_float(rn), _p(lines, C.c_float), C.c_int(lines.shape[0]), C.c_int(Lmax), _p(path, C.c_float),
           C.byref(ln), C.byref(cost))
    return dict(status=st, len=ln.value, cost=cost.value, path=path[:min(ln.value, Lmax)])


def smooth_one(wp, vmax=1.0, amax=0.5, dt=0.02, N=100, nsub=100):
    """The reference's post-planner sequence for ONE path (examples/zmq_test.cpp:66-93): from_path -> arclength ->
    gen_vel_prof<1> along the arclength (TOPP-RA + sampling) -> resample(nudge) -> curvature; wp float32 [n, 2]."""
    c = bezier_from_path(wp)
    tot, cum = bezier_arclength(c, nsub)
    cum = cum.astype(np.float32)
    AL = np.float32(0.0)
    for v in cum[:, -1]:
        AL = np.float32(AL + v)
    r = toppra([0.0], [float(AL)], [0.0], [0.0], [-vmax], [vmax], [-amax], [amax], N=N)
    s = toppra_sample([0.0], [float(AL)], [0.0], [0.0], r["x"], r["t"], float(np.float32(dt)))
    L = int(s["length"])
    pos = s["pos"][0, :L].astype(np.float32).copy()
    vel = s["vel"][0, :L].astype(np.float32)
    rs = bezier_resample(c, cum, AL, pos, True)
    return dict(ctrl=c, arclength=AL, length=L, pos=rs["pos"], vel=vel, pts=rs["pts"], curvature=rs["curvature"], seg=rs["seg"],
                status=int(rs["status"]), toppra_status=int(r["status"]))
