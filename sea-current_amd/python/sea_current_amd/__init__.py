"""sea_current_amd -- Python plumbing over libsea_current_hip.so (MI355X / gfx950).

The product is the C-ABI library (include/sea_current_hip.h) and the C++ header
sea-current_amd/sea_current.hpp; this module only binds the C ABI with ctypes so that
tests/ and bench.py can drive it with torch device tensors.  There is NO CPU fallback:
if the HIP library is missing or no GPU is present, calls raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
NATIVE_DIR = os.path.normpath(os.path.join(_PKG, "..", ".."))          # sea-current_amd/
REPO_ROOT = os.path.normpath(os.path.join(NATIVE_DIR, ".."))
LIB_PATH = os.environ.get("SC_LIB_PATH") or os.path.join(NATIVE_DIR, "libsea_current_hip.so")   # SC_LIB_PATH: experiment builds
HEADER_PATH = os.path.join(REPO_ROOT, "include", "sea_current_hip.h")

EDT_INF = 2**31 - 1
Q_OK, Q_NO_PATH, Q_BAD_ENDPOINT, Q_TRUNCATED, Q_RING_OVERFLOW = 0, 1, 2, 3, 4
K_EDT_COLBITS, K_EDT_BAND, K_MOVES, K_ASTAR, K_TOPPRA, K_TOPPRA_SAMPLE, K_BEZIER, K_ARCLENGTH, K_RESAMPLE, K_OCC, K_NEAREST, K_FMT, K_GATHER = range(13)

_lib = None


class SeaCurrentError(RuntimeError):
    pass


def build(force=False):
    """Compile libsea_current_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", NATIVE_DIR, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", NATIVE_DIR])
    return LIB_PATH


_vp, _i, _d, _i64p = C.c_void_p, C.c_int, C.c_double, C.POINTER(C.c_int64)
_SIGNATURES = {
    "sc_abi_version": (C.c_int, []),
    "sc_status_string": (C.c_char_p, [_i]),
    "sc_last_error": (C.c_char_p, [_vp]),
    "sc_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "sc_ctx_destroy": (_i, [_vp]),
    "sc_ctx_set_stream": (_i, [_vp, _vp]),
    "sc_ctx_use_own_stream": (_i, [_vp]),
    "sc_ctx_synchronize": (_i, [_vp]),
    "sc_ctx_set_timing": (_i, [_vp, _i]),
    "sc_ctx_reset_timing": (_i, [_vp]),
    "sc_ctx_get_timing": (_i, [_vp, _i, C.POINTER(_d), _i64p]),
    "sc_ctx_scratch_bytes": (_i, [_vp, _i64p]),
    "sc_edt_u8_i32": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sc_edt_u8_i32_host": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sc_moves_i32_u8": (_i, [_vp, _vp, _i, _i, C.c_int32, _vp]),
    "sc_astar_batch": (_i, [_vp, _vp, _i, _i, C.c_int32, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "sc_astar_batch_multi": (_i, [_vp, _vp, _i, _vp, _i, _i, C.c_int32, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "sc_astar_batch_host": (_i, [_vp, _vp, _i, _i, C.c_int32, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "sc_astar_last_expansions": (_i, [_vp, _i64p]),
    "sc_astar_debug_stats": (_i, [_vp, _vp, _i]),
    "sc_astar_debug_peek": (_i, [_vp, _vp]),
    "sc_astar_gfield": (_i, [_vp, _vp, _i, _i, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "sc_toppra_hermite_batch": (_i, [_vp, _i, _i, _i] + [_vp] * 6 + [_i, _vp, _vp, _d, _d] + [_vp] * 5),
    "sc_toppra_hermite_batch_host": (_i, [_vp, _i, _i, _i] + [_vp] * 6 + [_i, _vp, _vp, _d, _d] + [_vp] * 5),
    "sc_toppra_sample_batch": (_i, [_vp, _i, _i, _i] + [_vp] * 6 + [_d, _i] + [_vp] * 5),
    "sc_toppra_sample_batch_host": (_i, [_vp, _i, _i, _i] + [_vp] * 6 + [_d, _i] + [_vp] * 5),
    "sc_fmt_star_batch": (_i, [_vp, _vp, _i, _vp, _vp, _i, C.c_float, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "sc_fmt_star_batch_host": (_i, [_vp, _vp, _i, _vp, _vp, _i, C.c_float, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "sc_edt_nearest_i32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "sc_occ_from_rects": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sc_bezier_from_path_batch": (_i, [_vp, _vp, _vp, _i, _i, C.c_float, _vp, _i, _vp]),
    "sc_bezier_from_path_batch_host": (_i, [_vp, _vp, _vp, _i, _i, C.c_float, _vp, _i, _vp]),
    "sc_bezier_shrink_tangent_batch": (_i, [_vp, _vp, _vp, _i, C.c_float, _vp, _i, _vp]),
    "sc_bezier_shrink_tangent_batch_host": (_i, [_vp, _vp, _vp, _i, C.c_float, _vp, _i, _vp]),
    "sc_bezier_eval_batch": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "sc_bezier_eval_batch_host": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "sc_bezier_curve_batch": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp]),
    "sc_bezier_curve_batch_host": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "sc_chebfit_batch": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "sc_chebfit_batch_host": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "sc_chebeval_batch": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "sc_chebeval_batch_host": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "sc_bezier_arclength_batch_host": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "sc_bezier_arclength_batch": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "sc_bezier_resample_batch": (_i, [_vp] * 5 + [_i, _i, _i, _vp, _vp, _i] + [_vp] * 5),
    "sc_bezier_resample_batch_host": (_i, [_vp] * 5 + [_i, _i, _i, _vp, _vp, _i] + [_vp] * 5),
    "sc_rank_range": (None, [_i, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "sc_comm_unique_id": (_i, [_vp]),
    "sc_comm_init": (_i, [_vp, _vp, _i, _i]),
    "sc_comm_adopt": (_i, [_vp, _vp, _i, _i]),
    "sc_comm_destroy": (_i, [_vp]),
    "sc_allgather_paths": (_i, [_vp] * 5 + [_i, _i, _i, _i] + [_vp] * 5 + [C.c_int64, _vp, _vp]),
    "sc_allgather_last_bytes": (_i, [_vp, _i64p]),
    "sc_gather_msg_words": (C.c_int64, [_i, _i, _i]),
    "sc_gather_pack": (_i, [_vp] * 5 + [_i] * 6 + [_vp]),
    "sc_gather_unpack": (_i, [_vp, _vp, _i, _i, _i, _i] + [_vp] * 5 + [C.c_int64, _vp, _vp]),
}
EXPORTS = tuple(_SIGNATURES)


def lib():
    """Load the in-tree HIP library; fail loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SeaCurrentError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                  "or `make -C sea-current_amd` (there is no CPU fallback)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def _ptr(t):
    """Device/host pointer of a torch tensor or numpy array (must be contiguous)."""
    if t is None:
        return None
    if isinstance(t, np.ndarray):
        assert t.flags["C_CONTIGUOUS"]
        return t.ctypes.data
    assert t.is_contiguous()
    return t.data_ptr()


class Context:
    """One sc_ctx: one GPU, one stream.  `device` is the HIP device ordinal."""

    def __init__(self, device=0, use_torch_stream=True):
        # torch brings its own HIP runtime: let it initialise first, so that the library binds to the runtime that is already
        # in the process (loaded the other way round, the second runtime finds no GPU)
        import torch
        torch.cuda.is_available()
        self._l = lib()
        h = C.c_void_p()
        st = self._l.sc_ctx_create(device, C.byref(h))
        if st != 0:
            raise SeaCurrentError(f"sc_ctx_create(device={device}): {self._l.sc_status_string(st).decode()} "
                                  "(a gfx950 GPU is required; there is no CPU fallback)")
        self._h = h
        self.device = device
        if use_torch_stream:
            import torch
            self.set_stream(torch.cuda.current_stream(device).cuda_stream)

    def _ck(self, st, what):
        if st != 0:
            raise SeaCurrentError(f"{what}: {self._l.sc_status_string(st).decode()}: "
                                  f"{self._l.sc_last_error(self._h).decode()}")

    def close(self):
        if self._h:
            self._l.sc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        self._ck(self._l.sc_ctx_set_stream(self._h, C.c_void_p(hip_stream or None)), "sc_ctx_set_stream")

    def use_own_stream(self):
        self._ck(self._l.sc_ctx_use_own_stream(self._h), "sc_ctx_use_own_stream")

    def synchronize(self):
        self._ck(self._l.sc_ctx_synchronize(self._h), "sc_ctx_synchronize")

    # -- timing
    def set_timing(self, on):
        self._ck(self._l.sc_ctx_set_timing(self._h, int(bool(on))), "sc_ctx_set_timing")

    def reset_timing(self):
        self._ck(self._l.sc_ctx_reset_timing(self._h), "sc_ctx_reset_timing")

    def get_timing(self, kid):
        ms, n = C.c_double(0), C.c_int64(0)
        self._ck(self._l.sc_ctx_get_timing(self._h, kid, C.byref(ms), C.byref(n)), "sc_ctx_get_timing")
        return ms.value, n.value

    def scratch_bytes(self):
        b = C.c_int64(0)
        self._ck(self._l.sc_ctx_scratch_bytes(self._h, C.byref(b)), "sc_ctx_scratch_bytes")
        return b.value

    # -- device-pointer entry points (torch CUDA tensors)
    def edt(self, occ, out=None):
        """occ: uint8 [B,H,W] or [H,W] on the GPU -> int32 d2 of the same shape."""
        import torch
        assert occ.is_cuda and occ.dtype == torch.uint8
        o3 = occ if occ.dim() == 3 else occ.unsqueeze(0)
        B, H, W = o3.shape
        if out is None:
            out = torch.empty(o3.shape, dtype=torch.int32, device=occ.device)
        self._ck(self._l.sc_edt_u8_i32(self._h, _ptr(o3.contiguous()), W, H, B, _ptr(out)), "sc_edt_u8_i32")
        return out if occ.dim() == 3 else out.view(H, W)

    def moves(self, d2, r2=0):
        import torch
        H, W = d2.shape
        out = torch.empty((H, W), dtype=torch.uint8, device=d2.device)
        self._ck(self._l.sc_moves_i32_u8(self._h, _ptr(d2), W, H, r2, _ptr(out)), "sc_moves_i32_u8")
        return out

    def astar_batch(self, d2, start, goal, r2=0, Lmax=4096, out=None):
        """d2 int32 [H,W]; start/goal int32 [Q] (GPU).  Returns dict of GPU tensors."""
        import torch
        H, W = d2.shape
        Q = start.shape[0]
        if out is None:
            out = dict(path=torch.empty((Q, Lmax), dtype=torch.int32, device=d2.device),
                       len=torch.empty(Q, dtype=torch.int32, device=d2.device),
                       cost=torch.empty(Q, dtype=torch.int32, device=d2.device),
                       status=torch.empty(Q, dtype=torch.int32, device=d2.device))
        self._ck(self._l.sc_astar_batch(self._h, _ptr(d2), W, H, r2, _ptr(start), _ptr(goal), Q, Lmax,
                                        _ptr(out["path"]), _ptr(out["len"]), _ptr(out["cost"]), _ptr(out["status"])),
                 "sc_astar_batch")
        return out

    def astar_batch_multi(self, d2, qgrid, start, goal, r2=0, Lmax=4096, out=None):
        """Several grids in one launch: d2 int32 [G,H,W], qgrid int32 [Q] (grid of every query), start/goal int32 [Q]."""
        import torch
        G, H, W = d2.shape
        Q = start.shape[0]
        dev = d2.device
        if out is None:
            out = dict(path=torch.empty((Q, Lmax), dtype=torch.int32, device=dev), len=torch.empty(Q, dtype=torch.int32, device=dev),
                       cost=torch.empty(Q, dtype=torch.int32, device=dev), status=torch.empty(Q, dtype=torch.int32, device=dev))
        self._ck(self._l.sc_astar_batch_multi(self._h, _ptr(d2), G, _ptr(qgrid), W, H, r2, _ptr(start), _ptr(goal), Q, Lmax,
                                              _ptr(out["path"]), _ptr(out["len"]), _ptr(out["cost"]), _ptr(out["status"])), "sc_astar_batch_multi")
        return out

    # ---- multi-GPU gather (RCCL through the C ABI) ----
    @staticmethod
    def comm_unique_id():
        """128 bytes of an ncclUniqueId (call on one rank, hand to all)."""
        buf = C.create_string_buffer(128)
        st = lib().sc_comm_unique_id(buf)
        if st != 0:
            raise SeaCurrentError(f"sc_comm_unique_id: {lib().sc_status_string(st).decode()}")
        return buf.raw

    def comm_init(self, unique_id, world, rank):
        self._ck(self._l.sc_comm_init(self._h, C.c_char_p(unique_id), world, rank), "sc_comm_init")
        self.world, self.rank = world, rank

    def allgather_paths(self, out, Q_total, cap_cells, want_path=False, bufs=None):
        """Every rank's astar_batch results on every rank, in query order (sc_allgather_paths).  `out` = this rank's dict;
        returns dict(len, cost, status [Q_total], offsets int64 [Q_total+1], cells [world*cap_cells], truncated [1], path?)."""
        import torch
        dev = out["len"].device
        Ql, Lmax = out["path"].shape
        world = getattr(self, "world", 1)
        if bufs is None:
            bufs = dict(len=torch.empty(Q_total, dtype=torch.int32, device=dev), cost=torch.empty(Q_total, dtype=torch.int32, device=dev),
                        status=torch.empty(Q_total, dtype=torch.int32, device=dev),
                        offsets=torch.empty(Q_total + 1, dtype=torch.int64, device=dev),
                        cells=torch.empty(world * cap_cells, dtype=torch.int32, device=dev),
                        truncated=torch.zeros(1, dtype=torch.int32, device=dev))
            if want_path:
                bufs["path"] = torch.empty((Q_total, Lmax), dtype=torch.int32, device=dev)
        self._ck(self._l.sc_allgather_paths(self._h, _ptr(out["path"]), _ptr(out["len"]), _ptr(out["cost"]), _ptr(out["status"]), Ql, Q_total, Lmax,
                                            cap_cells, _ptr(bufs["len"]), _ptr(bufs["cost"]), _ptr(bufs["status"]), _ptr(bufs["offsets"]),
                                            _ptr(bufs["cells"]), bufs["cells"].numel(), _ptr(bufs.get("path")), _ptr(bufs["truncated"])),
                 "sc_allgather_paths")
        return bufs

    def gather_pack(self, out, Q_total, world, rank, cap_cells, msg=None, Lmax=None):
        """This rank's message of the gather (sc_gather_pack): int32 [sc_gather_msg_words].  `out` = the rank's astar_batch
        dict (may hold zero queries)."""
        import torch
        Ql = int(out["len"].shape[0])
        if Lmax is None:
            Lmax = int(out["path"].shape[1])
        words = int(self._l.sc_gather_msg_words(Q_total, world, cap_cells))
        if msg is None:
            msg = torch.empty(words, dtype=torch.int32, device=out["len"].device)
        assert msg.numel() == words
        nz = Ql > 0
        self._ck(self._l.sc_gather_pack(self._h, _ptr(out["path"]) if nz else None, _ptr(out["len"]) if nz else None,
                                        _ptr(out["cost"]) if nz else None, _ptr(out["status"]) if nz else None, Ql, Q_total, world, rank,
                                        Lmax, cap_cells, _ptr(msg)), "sc_gather_pack")
        return msg

    def gather_unpack(self, msgs, world, Q_total, Lmax, cap_cells, want_path=False, cells_capacity=None):
        """sc_gather_unpack: `msgs` = the world messages back to back in rank order -> the dict allgather_paths returns."""
        import torch
        dev = msgs.device
        cc = world * cap_cells if cells_capacity is None else cells_capacity
        bufs = dict(len=torch.empty(Q_total, dtype=torch.int32, device=dev), cost=torch.empty(Q_total, dtype=torch.int32, device=dev),
                    status=torch.empty(Q_total, dtype=torch.int32, device=dev), offsets=torch.empty(Q_total + 1, dtype=torch.int64, device=dev),
                    cells=torch.full((cc,), -7, dtype=torch.int32, device=dev), truncated=torch.zeros(1, dtype=torch.int32, device=dev))
        if want_path:
            bufs["path"] = torch.full((Q_total, Lmax), -7, dtype=torch.int32, device=dev)
        self._ck(self._l.sc_gather_unpack(self._h, _ptr(msgs), world, Q_total, Lmax, cap_cells, _ptr(bufs["len"]), _ptr(bufs["cost"]),
                                          _ptr(bufs["status"]), _ptr(bufs["offsets"]), _ptr(bufs["cells"]), cc, _ptr(bufs.get("path")),
                                          _ptr(bufs["truncated"])), "sc_gather_unpack")
        return bufs

    def allgather_last_bytes(self):
        n = C.c_int64()
        self._ck(self._l.sc_allgather_last_bytes(self._h, C.byref(n)), "sc_allgather_last_bytes")
        return n.value

    def astar_last_expansions(self):
        n = C.c_int64(0)
        self._ck(self._l.sc_astar_last_expansions(self._h, C.byref(n)), "sc_astar_last_expansions")
        return n.value

    def astar_debug_stats(self, Q):
        """Per-query (expansions, popped entries, kilo-cycles, steps) of the last astar_batch (synchronises)."""
        out = np.zeros((4, Q), dtype=np.int32)
        self._ck(self._l.sc_astar_debug_stats(self._h, _ptr(out), Q), "sc_astar_debug_stats")
        return out[0], out[1], out[2], out[3]

    def astar_gfield(self, d2, start, goal, r2=0):
        import torch
        H, W = d2.shape
        g = torch.empty((H, W), dtype=torch.int32, device=d2.device)
        cs = torch.empty(2, dtype=torch.int32, device=d2.device)
        self._ck(self._l.sc_astar_gfield(self._h, _ptr(d2), W, H, r2, int(start), int(goal), _ptr(g),
                                         cs.data_ptr(), cs.data_ptr() + 4), "sc_astar_gfield")
        self.synchronize()
        cost, status = cs.cpu().tolist()
        if status == Q_TRUNCATED:  # the path itself is not requested (Lmax = 1)
            status = Q_OK
        return g.cpu().numpy().view(np.uint32), cost, status

    def toppra(self, p0, p1, v0, v1, vlim_lo, vlim_hi, alim_lo, alim_hi, N=100, sd_start=0.0, sd_end=0.0):
        """All inputs float64 GPU tensors [P,dof] (vlim may be [P,N+1,dof])."""
        import torch
        P, dof = p0.shape
        per_stage = int(vlim_lo.dim() == 3)
        dev = p0.device
        out = dict(K=torch.empty((P, N + 1, 2), dtype=torch.float64, device=dev),
                   x=torch.empty((P, N + 1), dtype=torch.float64, device=dev),
                   u=torch.empty((P, N), dtype=torch.float64, device=dev),
                   t=torch.empty((P, N + 1), dtype=torch.float64, device=dev),
                   status=torch.empty(P, dtype=torch.int32, device=dev))
        self._ck(self._l.sc_toppra_hermite_batch(self._h, P, dof, N, _ptr(p0), _ptr(p1), _ptr(v0), _ptr(v1),
                                                 _ptr(vlim_lo), _ptr(vlim_hi), per_stage, _ptr(alim_lo), _ptr(alim_hi),
                                                 sd_start, sd_end, _ptr(out["K"]), _ptr(out["x"]), _ptr(out["u"]),
                                                 _ptr(out["t"]), _ptr(out["status"])), "sc_toppra_hermite_batch")
        return out

    def toppra_sample(self, p0, p1, v0, v1, x, t, dt, max_len):
        import torch
        P, dof = p0.shape
        N = x.shape[1] - 1
        dev = p0.device
        out = dict(pos=torch.zeros((P, dof, max_len), dtype=torch.float32, device=dev),
                   vel=torch.zeros((P, dof, max_len), dtype=torch.float32, device=dev),
                   acc=torch.zeros((P, dof, max_len), dtype=torch.float32, device=dev),
                   time=torch.zeros((P, max_len), dtype=torch.float64, device=dev),
                   length=torch.empty(P, dtype=torch.int32, device=dev))
        self._ck(self._l.sc_toppra_sample_batch(self._h, P, dof, N, _ptr(p0), _ptr(p1), _ptr(v0), _ptr(v1), _ptr(x),
                                                _ptr(t), float(dt), max_len, _ptr(out["pos"]), _ptr(out["vel"]),
                                                _ptr(out["acc"]), _ptr(out["time"]), _ptr(out["length"])),
                 "sc_toppra_sample_batch")
        return out

    def fmt_star(self, samples, starts, goals, rn, lines, Lmax=256):
        """FMT* (the reference's planner) for Q queries over shared samples.  GPU float32 tensors: samples [n,2], starts / goals
        [Q,2], lines [E,4] -> dict(path [Q,Lmax,2], len, cost, status)."""
        import torch
        Q = starts.shape[0]
        dev = starts.device
        out = dict(path=torch.zeros((Q, Lmax, 2), dtype=torch.float32, device=dev), len=torch.zeros(Q, dtype=torch.int32, device=dev),
                   cost=torch.zeros(Q, dtype=torch.float32, device=dev), status=torch.zeros(Q, dtype=torch.int32, device=dev))
        E = 0 if lines is None else lines.shape[0]
        self._ck(self._l.sc_fmt_star_batch(self._h, _ptr(samples), samples.shape[0], _ptr(starts), _ptr(goals), Q, float(rn),
                                           _ptr(lines) if E else None, E, Lmax, _ptr(out["path"]), _ptr(out["len"]), _ptr(out["cost"]),
                                           _ptr(out["status"])), "sc_fmt_star_batch")
        return out

    def edt_nearest(self, occ, d2):
        """occ uint8 [B,H,W] or [H,W] and its d2 (GPU) -> int32 index of the nearest occupied cell per cell (-1: none)."""
        import torch
        o3 = occ if occ.dim() == 3 else occ[None]
        B, H, W = o3.shape
        out = torch.empty((B, H, W), dtype=torch.int32, device=occ.device)
        self._ck(self._l.sc_edt_nearest_i32(self._h, _ptr(o3), _ptr(d2), W, H, B, _ptr(out)), "sc_edt_nearest_i32")
        return out if occ.dim() == 3 else out[0]

    def occ_from_rects(self, rects, W, H, base=None, free_border=True, out=None):
        """rects int32 [R,4] (x0,y0,x1,y1 exclusive; GPU) painted over `base` uint8 [H,W] (or an empty grid) -> occ uint8 [H,W]."""
        import torch
        occ = out if out is not None else torch.empty((H, W), dtype=torch.uint8, device=rects.device)
        self._ck(self._l.sc_occ_from_rects(self._h, _ptr(base) if base is not None else None, _ptr(rects), rects.shape[0], W, H,
                                           1 if free_border else 0, _ptr(occ)), "sc_occ_from_rects")
        return occ

    def bezier_from_path(self, path, npts, start_angle=float("nan"), lines=None):
        """path float32 [P,n_max,2], npts int32 [P] (GPU) -> ctrl float32 [P,n_max-1,4,2]."""
        import torch
        P, n_max, _ = path.shape
        ctrl = torch.empty((P, n_max - 1, 4, 2), dtype=torch.float32, device=path.device)
        nl = 0 if lines is None else lines.shape[0]
        self._ck(self._l.sc_bezier_from_path_batch(self._h, _ptr(path), _ptr(npts), P, n_max, start_angle,
                                                   _ptr(lines) if nl else None, nl, _ptr(ctrl)), "sc_bezier_from_path_batch")
        return ctrl

    def bezier_eval(self, ctrl, seg, t, order=0):
        import torch
        M = t.shape[0]
        out = torch.empty((M, 2), dtype=torch.float32, device=t.device)
        self._ck(self._l.sc_bezier_eval_batch(self._h, _ptr(ctrl), _ptr(seg), _ptr(t), M, order, _ptr(out)), "sc_bezier_eval_batch")
        return out

    def bezier_curve(self, ctrl, seg, t):
        """General degree: ctrl float32 [S, degree+1, 2] (GPU), seg int32 [M], t float32 [M] -> points float32 [M, 2]."""
        import torch
        M = t.shape[0]
        out = torch.empty((M, 2), dtype=torch.float32, device=t.device)
        self._ck(self._l.sc_bezier_curve_batch(self._h, _ptr(ctrl), ctrl.shape[1] - 1, _ptr(seg), _ptr(t), M, _ptr(out)), "sc_bezier_curve_batch")
        return out

    def chebfit(self, x, y, off, degree):
        """B least-squares Chebyshev fits; x, y float32 [total], off int32 [B+1] (all GPU) -> (coef [B, degree], xrange [B, 2])."""
        import torch
        B = off.shape[0] - 1
        coef = torch.empty((B, degree), dtype=torch.float32, device=x.device)
        xr = torch.empty((B, 2), dtype=torch.float32, device=x.device)
        self._ck(self._l.sc_chebfit_batch(self._h, _ptr(x), _ptr(y), _ptr(off), B, x.shape[0], degree, _ptr(coef), _ptr(xr)), "sc_chebfit_batch")
        return coef, xr

    def chebeval(self, x, off, coef, xr):
        import torch
        B, degree = coef.shape
        y = torch.empty_like(x)
        self._ck(self._l.sc_chebeval_batch(self._h, _ptr(x), _ptr(off), B, degree, _ptr(coef), _ptr(xr), _ptr(y)), "sc_chebeval_batch")
        return y

    def bezier_arclength(self, ctrl, nsub=100):
        """ctrl float32 [...,4,2] (GPU) -> (cum float32 [S,nsub+1], seg_len float32 [S])."""
        import torch
        c = ctrl.reshape(-1, 4, 2)
        S = c.shape[0]
        cum = torch.empty((S, nsub + 1), dtype=torch.float32, device=ctrl.device)
        seg_len = torch.empty(S, dtype=torch.float32, device=ctrl.device)
        self._ck(self._l.sc_bezier_arclength_batch(self._h, _ptr(c), S, nsub, _ptr(cum), _ptr(seg_len)), "sc_bezier_arclength_batch")
        return cum, seg_len

    def bezier_resample(self, ctrl, cum, arclength, seg_off, profile_pos, prof_off, nudge=True, want_curvature=True):
        """B splines (GPU tensors): ctrl float32 [S,4,2], cum float32 [S,nsub+1], arclength float32 [B], seg_off int32 [B+1],
        profile_pos float32 [M] (nudged IN PLACE when `nudge`), prof_off int32 [B+1] -> dict(pts [M,2], t [M], seg [M],
        curvature [M], status [B])."""
        import torch
        S, m = cum.shape
        B = arclength.shape[0]
        M = profile_pos.shape[0]
        dev = profile_pos.device
        out = dict(pts=torch.zeros((M, 2), dtype=torch.float32, device=dev), t=torch.zeros(M, dtype=torch.float32, device=dev),
                   seg=torch.zeros(M, dtype=torch.int32, device=dev), status=torch.zeros(B, dtype=torch.int32, device=dev),
                   curvature=torch.zeros(M, dtype=torch.float32, device=dev) if want_curvature else None)
        self._ck(self._l.sc_bezier_resample_batch(self._h, _ptr(ctrl), _ptr(cum), _ptr(arclength), _ptr(seg_off), B, S, m - 1,
                                                  _ptr(profile_pos), _ptr(prof_off), 1 if nudge else 0, _ptr(out["pts"]), _ptr(out["t"]),
                                                  _ptr(out["seg"]), _ptr(out["curvature"]) if want_curvature else None,
                                                  _ptr(out["status"])), "sc_bezier_resample_batch")
        return out

    # -- host-pointer entry points (numpy)
    def edt_host(self, occ):
        occ = np.ascontiguousarray(occ, dtype=np.uint8)
        o3 = occ if occ.ndim == 3 else occ[None]
        B, H, W = o3.shape
        d2 = np.empty(o3.shape, dtype=np.int32)
        self._ck(self._l.sc_edt_u8_i32_host(self._h, _ptr(o3), W, H, B, _ptr(d2)), "sc_edt_u8_i32_host")
        return d2 if occ.ndim == 3 else d2[0]

    def astar_batch_host(self, d2, start, goal, r2=0, Lmax=4096):
        d2 = np.ascontiguousarray(d2, dtype=np.int32)
        start = np.ascontiguousarray(start, dtype=np.int32)
        goal = np.ascontiguousarray(goal, dtype=np.int32)
        H, W = d2.shape
        Q = start.shape[0]
        out = dict(path=np.full((Q, Lmax), -1, dtype=np.int32), len=np.zeros(Q, np.int32),
                   cost=np.zeros(Q, np.int32), status=np.zeros(Q, np.int32))
        self._ck(self._l.sc_astar_batch_host(self._h, _ptr(d2), W, H, r2, _ptr(start), _ptr(goal), Q, Lmax,
                                             _ptr(out["path"]), _ptr(out["len"]), _ptr(out["cost"]),
                                             _ptr(out["status"])), "sc_astar_batch_host")
        return out
