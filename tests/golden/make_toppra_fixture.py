#!/usr/bin/env python3
"""Generate tests/golden/toppra_1dof_output.npz from the reference's only recorded
output, /root/reference/examples/output.json (produced by examples/zmq_test.py:7-10
through examples/zmq_test.cpp:25-101).  Data only -- no reference source is copied.

Input of the recorded run (examples/zmq_test.py:7-10):
    acc -0.5/0.5, vel -0.25/0.25, waypoints (0,0),(10,0),(10,10)
The JSON holds float32 values printed as doubles; they are stored back as float32
(time: float32-rounded doubles) so the fixture is byte-faithful to the record.
Run in the build container only (the reference does not exist on the GPU box).
"""
import json
import os

import numpy as np

SRC = "/root/reference/examples/output.json"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "toppra_1dof_output.npz")

d = json.load(open(SRC))
f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32)
out = dict(
    acc_lim=np.array([-0.5, 0.5]), vel_lim=np.array([-0.25, 0.25]),
    waypoints=np.array([[0, 0], [10, 0], [10, 10]], dtype=np.float32),
    arclength=np.float32(d["arclength"]["arclength"]),
    arclength_segments=f32(d["arclength"]["segments"]),
    arclength_positions=f32(d["arclength"]["positions"]),
    pos=f32(d["pos"][0]), vel=f32(d["vel"][0]), acc=f32(d["acc"][0]),
    time=f32(d["time"]), pos_x=f32(d["pos_x"]), pos_y=f32(d["pos_y"]), ang_vel=f32(d["ang_vel"]),
)
for k in ("pos", "vel", "acc", "time", "pos_x", "pos_y", "ang_vel"):
    ref = np.asarray(d[k][0] if k in ("pos", "vel", "acc") else d[k], dtype=np.float64)
    assert np.array_equal(out[k].astype(np.float64), ref), k  # lossless: values were float32
np.savez_compressed(DST, **out)
print("wrote", DST, os.path.getsize(DST), "bytes")
