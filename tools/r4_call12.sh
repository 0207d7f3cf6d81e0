#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c12; mkdir -p $O
cd $R
for n in 1 2 3 1 2; do
python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --inflight $n > $O/b_inflight${n}_$RANDOM.json 2>> $O/err.txt
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/b_*.json')): d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],1), d['config']['clips_in_flight_per_gpu'], d['config']['launch'][:40])"
tail -5 $O/err.txt
