#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c16; mkdir -p $O
cd $R
for r in 1 2; do
python bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-profile --no-slot-wait > $O/b_nowait_$r.json 2>> $O/err.txt
python bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-profile > $O/b_wait_$r.json 2>> $O/err.txt
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/b_*.json')): d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],1), d['config']['host_cpu_ms_per_clip'], d['config']['host_wall_in_enqueue_calls_ms_per_clip'])"
