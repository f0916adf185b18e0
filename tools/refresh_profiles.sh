#!/bin/bash
# Run ON THE GPU BOX (through gpurun, from the repo root): collects everything profiles/ is derived from into
# gpurun_out/prof/.  Afterwards, in the build container: python tools/refresh_profiles.py r03
#   gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh'
set -e -o pipefail
R=$PWD
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
echo bench done
SC_BENCH_FORCE_DIST=1 timeout -k 5 300 python3 $R/bench.py --no-cpu-baseline --replan-frames 0 > $O/bench_forced_dist.json 2> $O/bench_forced_dist.err
echo forced-dist bench done
timeout -k 5 200 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --only-main-map --replan-frames 0 > $O/bench_steps20.json 2> $O/bench_steps20.err
echo 20-step bench done
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o b -- python3 $R/bench.py --steps 16 --warmup 16 --no-cpu-baseline --repeats 2 > $O/bench_stats.log 2>&1
echo bench stats done
for cfg in "1024 64 salt20" "1024 64 blocks" "1024 64 salt05" "4096 4 salt20" "4096 4 blocks" "4096 16 salt20"; do
    set -- $cfg
    timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/edt_stats_$1_$2_$3 -o e -- python3 $R/tools/edt_variants.py libsea_current_hip.so $1 $2 $3 > $O/edt_stats_$1_$2_$3.log 2>&1
    echo edt stats $cfg done
done
# HBM traffic of the two roofline workloads: FETCH_SIZE and WRITE_SIZE in separate passes
for cfg in "1024 64 salt20" "4096 4 salt20"; do
    set -- $cfg
    for ctr in FETCH_SIZE WRITE_SIZE; do
        timeout -k 5 120 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/edt_$1_$ctr -o e -- python3 $R/tools/edt_variants.py libsea_current_hip.so $1 $2 $3 > $O/edt_$1_$ctr.log 2>&1
        echo edt $1 $ctr done
    done
done
# what bounds the band kernels on block-type maps: VALU instructions, VALU-active cycles, wave cycles, busy cycles, GPU clock
for cfg in "1024 64 blocks" "4096 4 blocks" "1024 64 salt20" "4096 4 salt20"; do
    set -- $cfg
    for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
        tag=$(echo $grp | tr ' ' '_')
        timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/edtpmc_$1_$3_$tag -o e -- python3 $R/tools/edt_variants.py libsea_current_hip.so $1 $2 $3 > $O/edtpmc_$1_$3_$tag.log 2>&1
    done
    echo edt pmc $cfg done
done
AST="python3 $R/tools/astar_saturation.py salt20 6144"
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $grp | tr ' ' '_')
    timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/astar_$tag -o a -- $AST > $O/astar_$tag.log 2>&1
    echo astar pmc $grp done
done
# the longest search of the headline batch alone on the chip (the tail of a one-call batch)
timeout -k 5 120 python3 $R/tools/astar_longest_pmc.py 10 > $O/astar_longest.json 2> $O/astar_longest.err
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    tag=$(echo $grp | tr ' ' '_')
    timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/astarlong_$tag -o a -- python3 $R/tools/astar_longest_pmc.py 10 > $O/astarlong_$tag.log 2>&1
done
echo astar longest done
timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d $O/toppra_pmc -o t -- python3 $R/tools/toppra_one.py > $O/toppra_pmc.log 2>&1
echo toppra pmc done
timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/toppra_stats -o t -- python3 $R/tools/toppra_one.py > $O/toppra_stats.log 2>&1
echo toppra stats done
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/smoothing_stats -o s -- python3 $R/tools/smoothing_one.py > $O/smoothing_stats.log 2>&1
echo smoothing stats done
timeout -k 5 60 $R/tools/microbench/min3_mb.bin > $O/min3_mb.log 2>&1 || true
timeout -k 5 60 $R/tools/microbench/lds_mb.bin > $O/lds_mb.log 2>&1 || true
echo all done
