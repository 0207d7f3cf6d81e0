#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c13; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "in_flight or hipgraph" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for n in 2 3 4 6; do
python bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-profile --inflight $n > $O/b_inflight${n}.json 2>> $O/err.txt
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/b_*.json')): d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],1), d['config']['clips_in_flight_per_gpu'], d['config']['host_cpu_ms_per_clip'])"
