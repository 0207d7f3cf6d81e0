#!/bin/bash
# round-3 GPU call 2: fused FF ablations, attn40 stream cursors, new parity tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
( timeout 300 tools/cbench/cbench ff 65536 v=1,2,3,4,5,6,7,8,9 ) > $O/r3b_cbench_ff.log 2>&1
( timeout 300 tools/cbench/cbench attn-det; timeout 200 tools/cbench/cbench attn-time 1,0 ) > $O/r3b_cbench_attn.log 2>&1
timeout 900 python -m pytest tests/test_models_gpu.py -k "hipgraph" tests/test_ops_gpu.py -x -q > $O/r3b_pytest_a.log 2>&1
timeout 1200 python -m pytest tests/test_ops_large_gpu.py -k "row_counts or 768 or l0_two_segment" -x -q > $O/r3b_pytest_b.log 2>&1
timeout 1500 python -m pytest tests/test_full_size_gpu.py -k "cfg or 768 or ten_steps or fp8" -x -q > $O/r3b_pytest_c.log 2>&1
cp $O/parity_report.json $O/r3b_parity_report.json 2>/dev/null
tail -3 $O/r3b_cbench_ff.log $O/r3b_cbench_attn.log $O/r3b_pytest_a.log $O/r3b_pytest_b.log $O/r3b_pytest_c.log
