#!/bin/bash
# Run ON THE GPU BOX (through gpurun, from the repo root): collects everything profiles/ is derived from into
# gpurun_out/prof/.  Afterwards, in the build container: python tools/refresh_profiles.py
#   gpurun --timeout 900 -- 'bash tools/refresh_profiles.sh'
set -e -o pipefail
R=$PWD
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
EDT="python3 $R/tools/edt_variants.py libsea_current_hip.so 1024 64 salt20"
timeout -k 5 200 python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo bench done
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o b -- python3 $R/bench.py --steps 16 --warmup 16 --no-cpu-baseline > $O/bench_stats.log 2>&1
echo bench stats done
timeout -k 5 90 rocprofv3 --kernel-trace --stats --output-format csv -d $O/edt_stats -o e -- $EDT > $O/edt_stats.log 2>&1
echo edt stats done
timeout -k 5 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/edt_fetch -o e -- $EDT > $O/edt_fetch.log 2>&1
echo edt fetch done
timeout -k 5 90 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/edt_write -o e -- $EDT > $O/edt_write.log 2>&1
echo edt write done
