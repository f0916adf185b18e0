"""Error behaviour of the C ABI on a GPU box: bad arguments come back as status codes (nothing exits, nothing faults),
empty batches are no-ops, and a context survives a rejected call."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_invalid_arguments_are_status_codes():
    import torch
    import sea_current_amd as sc
    ctx = sc.Context(0)
    l, h = ctx._l, ctx._h
    occ = torch.zeros((8, 8), dtype=torch.uint8, device="cuda")
    d2 = torch.zeros((8, 8), dtype=torch.int32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    INVALID = 1
    assert l.sc_edt_u8_i32(h, None, 8, 8, 1, p(d2)) == INVALID
    assert l.sc_edt_u8_i32(h, p(occ), 0, 8, 1, p(d2)) == INVALID
    assert l.sc_edt_u8_i32(h, p(occ), 8, 1 << 20, 1, p(d2)) == INVALID          # beyond SC_MAX_DIM
    assert l.sc_edt_u8_i32(None, p(occ), 8, 8, 1, p(d2)) == INVALID
    q = torch.zeros(4, dtype=torch.int32, device="cuda")
    path = torch.zeros((4, 16), dtype=torch.int32, device="cuda")
    assert l.sc_astar_batch(h, p(d2), 8, 8, 0, p(q), p(q), 4, 0, p(path), p(q), p(q), p(q)) == INVALID   # Lmax 0
    assert l.sc_astar_batch(h, p(d2), 8, 8, 0, p(q), p(q), -1, 16, p(path), p(q), p(q), p(q)) == INVALID
    assert l.sc_astar_batch(h, p(d2), 8, 8, 0, p(q), p(q), 0, 16, p(path), p(q), p(q), p(q)) == 0          # empty batch: no-op
    assert l.sc_bezier_resample_batch(h, p(d2), p(d2), p(d2), p(q), 1, 1, 100000, p(d2), p(q), 0, None, None, None, None, p(q)) == INVALID
    assert l.sc_fmt_star_batch(h, p(d2), 5000, p(d2), p(d2), 1, C.c_float(1.0), None, 0, 8, p(d2), p(q), p(d2), p(q)) == INVALID
    assert l.sc_occ_from_rects(h, None, None, 3, 8, 8, 1, p(occ)) == INVALID                                  # R > 0 without rects
    assert l.sc_status_string(INVALID)
    # the context still works
    occ[3, 3] = 1
    out = ctx.edt(occ)
    torch.cuda.synchronize()
    assert int(out[3, 3]) == 0 and int(out[0, 0]) == 18
    # per-query statuses: out-of-range / blocked endpoints do not fail the batch
    res = ctx.astar_batch(out, torch.tensor([-5, 27, 0], dtype=torch.int32, device="cuda"), torch.tensor([1, 1, 64], dtype=torch.int32, device="cuda"), Lmax=16)
    torch.cuda.synchronize()
    assert res["status"].tolist() == [2, 2, 2]
    ctx.close()
