"""Long stress run: HIP EDT vs the exact CPU oracle on random shapes (odd widths and heights, wide rows, very sparse and
very dense grids, batches); any differing cell is fatal (bounded version in tests/test_gpu_stress.py).
python tools/edt_stress.py [rounds]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sea-current_amd", "python"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import sea_current_amd as sc
from oracle import oracle
import stress_cases as cases
oracle.build()
ctx = sc.Context(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(7)
cells = 0
for r in range(rounds):
    cells += cases.edt_round(ctx, oracle, rng, max_cells=40_000_000)
    if r % 20 == 0:
        print("round", r, "ok", flush=True)
print("stress ok:", cells, "cells")
