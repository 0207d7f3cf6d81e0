#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
( timeout 300 tools/cbench/cbench ff 65536 v=1,2,4,5,7,8; timeout 100 tools/cbench/cbench ff 73728 v=1; timeout 100 tools/cbench/cbench ff 131072 v=1 ; timeout 100 tools/cbench/cbench ff 65536 f16 v=1 ) > $O/r3c_cbench_ff.log 2>&1
timeout 600 python -m pytest tests/test_ops_gpu.py -k "ff320" -x -q > $O/r3c_pytest.log 2>&1
timeout 600 python tools/r3_fp_check.py > $O/r3c_fp_check.json 2> $O/r3c_fp_check.err
tail -4 $O/r3c_cbench_ff.log $O/r3c_pytest.log; head -c 1500 $O/r3c_fp_check.json
