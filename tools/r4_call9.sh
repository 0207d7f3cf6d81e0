#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c9; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops_large_gpu.py -x -q -m gpu > $O/pytest_ops.log 2>&1
tail -3 $O/pytest_ops.log
python bench.py --steps 6 --warmup 2 --shape-breakdown > $O/bench.json 2> $O/bench.err
cp gpurun_out/shape_breakdown.json $O/ 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config']['launch']); print(json.dumps(d['kernels'])[:900]); print(json.dumps(d['kernel_symbols'])[:1800])"
timeout 600 python -m pytest tests/test_models_gpu.py -x -q -m gpu > $O/pytest_models.log 2>&1
tail -3 $O/pytest_models.log
