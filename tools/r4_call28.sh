#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c28; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/err.txt
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['one_clip_at_a_time'], d['roofline']['frac'], d['config']['kernel_routing'][:30], d.get('speedup_vs_cpu'))
print({k:(v['ms'],v['launches']) for k,v in d['kernels'].items() if v['ms']>3})"
tail -2 $O/err.txt
