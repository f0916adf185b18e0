#!/bin/bash
# A/B of A* builds on one box: parity subset, then the main-map bench line with the default and capped resident counts.
set -e -o pipefail
O=gpurun_out/r3ab; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_astar.py tests/test_gpu_stress.py tests/test_gpu_replan.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for dual in ${DUALS:-99999 3072}; do
  for rep in 1 2; do
    SC_ASTAR_DUAL=$dual timeout -k 10 300 python3 bench.py --no-cpu-baseline --only-main-map --replan-frames 0 > $O/b_${dual}_$rep.json 2> $O/b_${dual}_$rep.err
    python3 - $O/b_${dual}_$rep.json "$dual" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("dual=%s"%sys.argv[2], d["value"], d.get("value_min"), d.get("value_max"), d.get("value_depth1"))
PY
  done
done
