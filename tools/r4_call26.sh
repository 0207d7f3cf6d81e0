#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c26; mkdir -p $O
cd $R
run() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-profile "$@" 2>> $O/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$*', round(d['value'],3))"; }
for r in 1 2; do
run
run --gemm-variant 5
run --gemm-variant 4
run --set-option conv_fast=0
done
