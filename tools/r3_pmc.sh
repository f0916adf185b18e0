#!/bin/bash
# usage: bash tools/r3_pmc.sh <outdir-tag> <lib> <W> <batch> <map>   -- PMC groups for the EDT kernels of one workload
set -e -o pipefail
R=$PWD; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
    tag=$(echo $grp | tr ' ' '_')
    timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$tag -o e -- python3 $R/tools/edt_variants.py $2 $3 $4 $5 > $O/pmc_$tag.log 2>&1 || echo fail $tag
done
