#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6_n
O=gpurun_out/r6_n
B="--no-cpu-baseline --no-configs2 --no-fp16-leg --no-serial-leg --no-profile --steps 12"
for r in 1 2 3; do
timeout 400 python bench.py $B > $O/inplace_$r.json 2> $O/inplace_$r.err; grep -o '"value": [0-9.]*' $O/inplace_$r.json | head -1 | sed 's/^/in place /'
timeout 400 python bench.py $B --materialize-skip-concat > $O/materialized_$r.json 2> $O/materialized_$r.err; grep -o '"value": [0-9.]*' $O/materialized_$r.json | head -1 | sed 's/^/materialized /'
done
