#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 ) > $O/r3h_pytest.log 2>&1
cp $O/parity_report.json $O/r3h_parity_report.json 2>/dev/null
tail -40 $O/r3h_pytest.log
bash tools/round_end_gpu.sh r3h 2>&1 | tail -12
