#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c27; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_multigpu_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py > $O/bench_default.json 2> $O/err.txt
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['clip_latency_ms'], d['roofline']['frac'], d['config']['kernel_routing'][:30], d['cpu_baseline']['value'], d.get('speedup_vs_cpu'))"
