#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c11; mkdir -p $O
cd $R
for i in 1 2; do
(cd tools/_r3_snapshot && python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $O/b_r3_$i.json 2> $O/err_r3.txt)
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --set-option gemm4=0 > $O/b_r4_g4off_$i.json 2> $O/err.txt
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/b_*.json')): d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],1))"
tail -3 $O/err_r3.txt
