#!/bin/bash
# round 4, second GPU call: vendor yardstick, clean baseline bench, the new graph tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c2; mkdir -p $O
cd $R
python bench.py --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err
timeout 900 python tools/vendor_ab.py > $O/vendor_ab.log 2>&1
cp gpurun_out/vendor_ab.json $O/ 2>/dev/null
timeout 600 python -m pytest tests/test_multigpu_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "rccl or reload or hipgraph" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config'])"
tail -5 $O/vendor_ab.log
