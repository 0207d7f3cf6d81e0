#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c10; mkdir -p $O
cd $R
for i in 1 2; do
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --set-option gemm4=0 > $O/b_g4off_$i.json 2> $O/err.txt
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $O/b_g4on_$i.json 2>> $O/err.txt
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/b_*.json')): d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],1))"
rocm-smi --showclocks --showpower 2>/dev/null | head -20
