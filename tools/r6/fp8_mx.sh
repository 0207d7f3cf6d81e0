#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q > gpurun_out/r6_fp8_tests.log 2>&1; echo "fp8 tests rc=$?"; tail -4 gpurun_out/r6_fp8_tests.log
timeout 300 python tools/fp8_mx_bench.py > gpurun_out/r6_fp8_mx_bench.log 2>&1; cat gpurun_out/r6_fp8_mx_bench.log | tail -8
timeout 400 python bench.py --fp8-proj --no-cpu-baseline --no-configs2 --no-fp16-leg --no-serial-leg --no-profile > gpurun_out/r6_bench_fp8.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/r6_bench_fp8.log | head -1
timeout 400 python bench.py --fp8-proj --set-option fp8_mx=0 --no-cpu-baseline --no-configs2 --no-fp16-leg --no-serial-leg --no-profile > gpurun_out/r6_bench_fp8_nonscaled.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/r6_bench_fp8_nonscaled.log | head -1
timeout 400 python bench.py --no-cpu-baseline --no-configs2 --no-fp16-leg --no-serial-leg --no-profile > gpurun_out/r6_bench_nofp8.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/r6_bench_nofp8.log | head -1
