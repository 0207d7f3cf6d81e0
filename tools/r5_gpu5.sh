#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
B="--no-cpu-baseline --no-serial-leg --no-configs2 --steps 9 --warmup 3"
for ps in 1 0 1 0; do timeout 300 python bench.py $B --set-option producer_stats=$ps > $O/r5_bench_ps${ps}_$RANDOM.json 2> $O/r5_bench_ps$ps.err; done
python tools/r5_ps_summary.py
timeout 300 python -m pytest tests/test_ops_large_gpu.py -x -q -k "gemm4" > $O/r5_g4.log 2>&1; tail -3 $O/r5_g4.log
