#!/bin/bash
# round 5: the whole GPU suite on the current tree (what the driver runs at round end)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r5_gpu_suite.log
tail -8 gpurun_out/r5_gpu_suite.log
cp gpurun_out/parity_report.json gpurun_out/r5_parity_report.json 2>/dev/null
