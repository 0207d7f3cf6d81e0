#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c24; mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest.log 2>&1
cp gpurun_out/parity_report.json $O/parity_report.json 2>/dev/null
tail -30 $O/pytest.log
