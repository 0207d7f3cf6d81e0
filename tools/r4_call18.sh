#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c18; mkdir -p $O
cd $R
run() { python bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-profile "$@" 2>> $O/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$*', round(d['value'],3))"; }
run
run --set-option ff_fused=1
run --set-option gn_fused=0
run --set-option gemm_rs=1
run --set-option gemm_rs=0
run --set-option attn_order=0
run --set-option attn_order=1
run --set-option xattn_tiled=0
run --set-option temporal_mfma=1
run --set-option gemm_variant=3
run
