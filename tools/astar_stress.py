"""Long stress run: batched HIP A* vs the CPU oracle on many random maps (sizes, densities, clearances, map families);
any mismatch in status / cost / len / path / expansion count is fatal.  The `-m gpu` suite runs a bounded version
(tests/test_gpu_stress.py).  python tools/astar_stress.py [rounds]"""
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
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(2024)
nq = 0
for r in range(rounds):
    nq += cases.astar_round(ctx, oracle, rng, Q=96 if r % 10 else 6000)      # now and then more queries than the launch has slots
    if r % 25 == 0:
        print("round", r, "ok", flush=True)
print("stress ok:", nq, "queries, expansion counts equal to the oracle's in every round")
