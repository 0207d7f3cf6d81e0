#!/bin/bash
# round-3 GPU call 1: fused feed-forward kernel (cbench), hipGraph A/B, targeted tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
( for a in "65536" "65536 f16" "73728" "16384" "24576" "131072" "65536 noln"; do timeout 120 tools/cbench/cbench ff $a; done ) > $O/r3a_cbench_ff.log 2>&1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -k "ff320 or hipgraph" -x -q > $O/r3a_pytest.log 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/r3a_bench_graph.log 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-graph > $O/r3a_bench_nograph.log 2>&1
tail -5 $O/r3a_cbench_ff.log $O/r3a_pytest.log; tail -c 600 $O/r3a_bench_graph.log; tail -c 600 $O/r3a_bench_nograph.log
