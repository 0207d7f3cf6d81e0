#!/bin/bash
# final tree: the driver's round-end sequence (pytest -m gpu, smoke, bench.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_final_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r5_final_suite.log
tail -4 gpurun_out/r5_final_suite.log
cp gpurun_out/parity_report.json gpurun_out/r5_parity_report.json 2>/dev/null
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 500 python bench.py > gpurun_out/r5_final_bench.json 2> gpurun_out/r5_final_bench.err
echo "bench rc=$?"; cut -c1-260 gpurun_out/r5_final_bench.json
