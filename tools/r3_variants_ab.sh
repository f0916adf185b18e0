#!/bin/bash
# Bench line (main map only) for the product library and every experiment build under sea-current_amd/variants/.
set -e -o pipefail
O=gpurun_out/r3var; mkdir -p $O
for lib in base $(ls sea-current_amd/variants/*.so 2>/dev/null); do
  name=$(basename $lib .so)
  for rep in 1 2; do
    if [ $lib = base ]; then unset SC_LIB_PATH; else export SC_LIB_PATH=$PWD/$lib; fi
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --only-main-map --replan-frames 0 > $O/b_${name}_$rep.json 2> $O/b_${name}_$rep.err
    python3 - $O/b_${name}_$rep.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-10s value %.0f (%.0f..%.0f) depth1 %.0f kcyc_max %s" % (sys.argv[2], d["value"], d["value_min"], d["value_max"], d["value_depth1"], d["per_query_depth1"]["kilocycles_max"]))
PY
  done
done
for lib in base $(ls sea-current_amd/variants/*.so 2>/dev/null); do
  if [ $lib = base ]; then unset SC_LIB_PATH; else export SC_LIB_PATH=$PWD/$lib; fi
  echo "$(basename $lib .so): $(timeout -k 10 120 python3 tools/astar_saturation.py salt20 6144 2>&1 | tail -1)"
done
