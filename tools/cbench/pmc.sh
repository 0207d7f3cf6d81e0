#!/bin/bash
# rocprofv3 --pmc passes over cbench launches (no torch: a pass costs seconds).  One counter group per pass, --kernel-trace only.
#   tools/cbench/pmc.sh <tag> <cbench args...>     -> gpurun_out/pmc_<tag>/<group>/
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
export CBENCH_QUICK=1 TMPDIR=/tmp
declare -A G
G[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
G[sq2]="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA"
G[fetch]="FETCH_SIZE"
G[hit]="TCC_HIT_sum TCC_MISS_sum"
G[write]="WRITE_SIZE GRBM_GUI_ACTIVE"
mkdir -p $R/gpurun_out/pmc_$TAG
for g in ${PMC_GROUPS:-sq1 sq2 fetch write hit}; do
  (cd /tmp && timeout 120 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/$g -- $R/tools/cbench/cbench "$@" > $R/gpurun_out/pmc_$TAG/$g.log 2>&1) || echo "pass $g failed"
done
python3 $R/tools/pmc_summarize.py $R/gpurun_out/pmc_$TAG > $R/gpurun_out/pmc_$TAG/summary.txt 2>&1
find $R/gpurun_out/pmc_$TAG -name "*.csv" -size +200k -delete
