"""Long run: grids of 1..11 cells a side, end points anywhere (also outside the grid): EDT and A* against the oracle
(bounded version in tests/test_gpu_stress.py)."""
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
rng = np.random.default_rng(3)
print("tiny grids ok:", sum(cases.tiny_grid_round(ctx, oracle, rng) for _ in range(400)), "queries")
