#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c14; mkdir -p $O
cd $R
for r in 1 2; do
python bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-profile --set-option gemm4=0 > $O/b_g4off_$r.json 2>> $O/err.txt
python bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-profile > $O/b_g4nk40_$r.json 2>> $O/err.txt
python bench.py --steps 9 --warmup 3 --no-cpu-baseline --no-profile --set-option gemm4_min_nk=20 > $O/b_g4nk20_$r.json 2>> $O/err.txt
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/b_*.json')): d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],1), d['config']['clips_in_flight_per_gpu'])"
