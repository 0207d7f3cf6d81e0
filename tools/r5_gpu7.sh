#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
python -c "
from hallo_amd import ops
for k,v in ops.THROUGHPUT_OPTIONS.items(): ops.set_option(k,v)
import runpy; runpy.run_path('tools/r5_parts_bench.py', run_name='__main__')" > $O/r5_parts_bench.log 2>&1
echo "rc=$?"; grep -v amdgpu.ids $O/r5_parts_bench.log | cut -c1-330
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "row_parts or geglu_fused_layernorm or test_gemm_fused_layernorm" > $O/r5_parts_ops.log 2>&1; tail -3 $O/r5_parts_ops.log
