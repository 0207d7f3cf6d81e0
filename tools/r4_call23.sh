#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c23; mkdir -p $O
cd $R
run() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-profile "$@" 2>> $O/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$*', round(d['value'],3), d['config']['kernel_routing'][:12])"; }
for r in 1 2; do
run
run --set-option gemm_stage_min_tiles=0
run --set-option gemm_stage_min_tiles=100000
run --set-option split_k=0
run --set-option gemm_stage_min_tiles=0 --set-option split_k=0
run --latency-routing
done
