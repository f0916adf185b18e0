#!/bin/bash
# Round-3 probe (run through gpurun from the repo root): GPU suite + EDT timings + PMC evidence for the band kernel.
set -e -o pipefail
R=$PWD
O=$R/gpurun_out/r3probe
rm -rf $O && mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for cfg in "1024 64 salt20" "1024 64 blocks" "4096 4 salt20" "4096 4 blocks"; do
    set -- $cfg
    timeout -k 5 120 python3 $R/tools/edt_variants.py libsea_current_hip.so $1 $2 $3 >> $O/edt_times.log 2>&1
done
cat $O/edt_times.log
for cfg in "1024 64 blocks" "1024 64 salt20" "4096 4 salt20"; do
    set -- $cfg
    for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
        tag=$(echo $grp | tr ' ' '_')
        timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$1_$3_$tag -o e -- python3 $R/tools/edt_variants.py libsea_current_hip.so $1 $2 $3 > $O/pmc_$1_$3_$tag.log 2>&1
        echo pmc $cfg $grp done
    done
done
