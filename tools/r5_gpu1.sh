#!/bin/bash
# round 5, GPU call 1: full-size in-flight identity test, default bench, host-restricted rehearsals
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
cat /sys/fs/cgroup/cpu.max > $O/r5_box.txt 2>&1; nproc >> $O/r5_box.txt; python -c "import os;print(len(os.sched_getaffinity(0)))" >> $O/r5_box.txt
timeout 900 python -m pytest tests/test_models_gpu.py -x -q -k "in_flight or hipgraph" > $O/r5_models_inflight.log 2>&1
echo "models rc=$?" >> $O/r5_models_inflight.log
tail -3 $O/r5_models_inflight.log
timeout 600 python bench.py > $O/r5_bench_default.json 2> $O/r5_bench_default.err
echo "bench rc=$?"; cut -c1-600 $O/r5_bench_default.json
B="--no-cpu-baseline --no-profile --no-serial-leg --steps 9 --warmup 3"
timeout 300 taskset -c 0-1 python bench.py $B > $O/r5_bench_taskset2.json 2> $O/r5_bench_taskset2.err
echo "taskset2 rc=$?"; cut -c1-300 $O/r5_bench_taskset2.json
timeout 300 taskset -c 0 python bench.py $B > $O/r5_bench_taskset1.json 2> $O/r5_bench_taskset1.err
echo "taskset1 rc=$?"; cut -c1-300 $O/r5_bench_taskset1.json
timeout 300 python bench.py $B > $O/r5_bench_free.json 2> $O/r5_bench_free.err
echo "free rc=$?"; cut -c1-300 $O/r5_bench_free.json
