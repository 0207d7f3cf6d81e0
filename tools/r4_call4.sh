#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c4; mkdir -p $O
cd $R
timeout 600 bash tools/cbench/g4.sh > $O/g4.txt 2>&1
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops_large_gpu.py -x -q -m gpu > $O/pytest_ops.log 2>&1
tail -3 $O/pytest_ops.log
cat $O/g4.txt
