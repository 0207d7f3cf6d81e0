#!/bin/bash
# the full GPU parity suite on the current tree (+ smoke), with the per-test report
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6_suite
O=gpurun_out/r6_suite
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q --durations=40 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - S )) s"; tail -60 $O/pytest_gpu.log
cp gpurun_out/parity_report.json $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
