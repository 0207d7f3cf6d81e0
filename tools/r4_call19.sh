#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c19; mkdir -p $O
cd $R
run() { python bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-profile "$@" 2>> $O/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$*', round(d['value'],3))"; }
for r in 1 2; do
run
run --set-option gemm_rs=0
run --set-option gemm_rs=0 --set-option ff_fused=1
run --set-option ff_fused=1
run --set-option gemm_rs=0 --inflight 1
run --inflight 1
done
