#!/bin/bash
# rocprofv3 --pmc passes over one of the single-launch drivers (tools/pmc_attn.py, tools/pmc_gemm2.py, ...), one counter
# group per pass (SQ has 8 slots; FETCH_SIZE / WRITE_SIZE need separate passes), --kernel-trace only (no other trace domain).
#   tools/pmc_passes.sh <tag> <script.py> [ENV=VAL ...]      -> gpurun_out/pmc_<tag>/<group>/...counter_collection.csv
set -u
TAG=$1; SCRIPT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
declare -A G
G[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
G[sq2]="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA"
G[fetch]="FETCH_SIZE"
G[hit]="TCC_HIT_sum TCC_MISS_sum"
G[write]="WRITE_SIZE GRBM_GUI_ACTIVE"
for g in ${PMC_GROUPS:-sq1 sq2 fetch write}; do
  timeout 300 rocprofv3 --pmc ${G[$g]} --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/$g -- python $R/$SCRIPT > $R/gpurun_out/pmc_$TAG/$g.log 2>&1 || echo "pass $g failed (see $g.log)"
done
