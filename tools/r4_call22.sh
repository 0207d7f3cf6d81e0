#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c22; mkdir -p $O
cd $R
B="--set-option gemm_rs=0 --set-option ff_fused=1 --set-option gemm4=0 --set-option gn_fused=0"
run() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-profile "$@" 2>> $O/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$*', round(d['value'],3))"; }
for r in 1 2; do
run $B
run $B --set-option split_k=0
run $B --set-option tok_attn=0
run $B --set-option xattn_tiled=0
run $B --set-option temporal_mfma=1
run $B --set-option attn40=0
run $B --inflight 4
done
