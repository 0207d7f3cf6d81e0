#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
PMC_GROUPS="sq1 sq2 hit" bash tools/cbench/pmc.sh ff320 ff 65536 v=1 > $O/r3d_pmc.log 2>&1
cat $O/pmc_ff320/summary.txt | grep -A30 "ff320" | head -60
timeout 300 python -m pytest tests/test_ops_gpu.py -k "temporal" -x -q > $O/r3d_pytest.log 2>&1; tail -3 $O/r3d_pytest.log
timeout 300 python tools/temporal_bench.py > $O/r3d_temporal.log 2>&1; tail -9 $O/r3d_temporal.log
