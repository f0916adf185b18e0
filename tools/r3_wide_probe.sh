#!/bin/bash
# wide-row EDT kernel: timings of the full kernel and of its two phases alone, then PMC counters
set -e -o pipefail
R=$PWD; O=$R/gpurun_out/r3wide; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in libsea_current_hip.so variants/lib_EDT_ABLATE_PHASE1.so variants/lib_EDT_ABLATE_ROWS.so; do
  for c in "4096 4 salt20" "4096 4 blocks"; do python3 $R/tools/edt_variants.py $lib $c 2>&1 | grep "{" ; done
done | tee $O/times.log
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
    tag=$(echo $grp | tr ' ' '_')
    timeout -k 5 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_4096_salt20_$tag -o e -- python3 $R/tools/edt_variants.py libsea_current_hip.so 4096 4 salt20 > $O/pmc_$tag.log 2>&1
done
echo done
