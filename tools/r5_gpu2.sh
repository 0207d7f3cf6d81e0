#!/bin/bash
# round 5, GPU call: cfg_split / overlap_decode tests (reduced width + the full-size CFG trajectory), default bench with the configs2 leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_models_gpu.py -x -q -k "cfg_split or sliding_window or end_to_end" > $O/r5_split_models.log 2>&1
echo "models rc=$?" >> $O/r5_split_models.log; tail -4 $O/r5_split_models.log
timeout 600 python bench.py > $O/r5_bench_configs2.json 2> $O/r5_bench_configs2.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_bench_configs2.json"))
print(d["value"], d.get("one_clip_at_a_time",{}).get("value"), d.get("inflight_identity",{}).get("identical"))
print(json.dumps(d.get("configs2"), indent=1))
PY
tail -3 $O/r5_bench_configs2.err
timeout 1200 python -m pytest tests/test_full_size_gpu.py -x -q -k "trajectory and pipeline40cfg and cfg_split" > $O/r5_split_full.log 2>&1
echo "full rc=$?" >> $O/r5_split_full.log; tail -4 $O/r5_split_full.log
