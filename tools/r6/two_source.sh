#!/bin/bash
# round 6: skip concatenation read in place (two-source GroupNorm + split 1x1 shortcut): tests + end-to-end
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6_m
O=gpurun_out/r6_m
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "groupnorm" > $O/t_gn.log 2>&1; echo "gn rc=$?"; tail -2 $O/t_gn.log
timeout 900 python -m pytest tests/test_models_gpu.py -x -q -k "referencenet or unet3d_forward or end_to_end or call_batch" > $O/t_models.log 2>&1; echo "models rc=$?"; tail -2 $O/t_models.log
timeout 1200 python -m pytest tests/test_full_size_gpu.py -x -q -k "referencenet or unet3d_forward or trajectory" > $O/t_full.log 2>&1; echo "full rc=$?"; tail -2 $O/t_full.log
for r in 1 2; do
timeout 400 python bench.py --no-cpu-baseline --no-configs2 --no-fp16-leg --no-serial-leg --steps 12 > $O/bench_$r.json 2> $O/bench_$r.err; grep -o '"value": [0-9.]*' $O/bench_$r.json | head -1
done
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_m/bench_2.json"))
print({k:(v["ms"],v["launches"]) for k,v in d["kernels"].items()}, d["kernel_ms_per_clip"], d["inflight_identity"]["identical"])
PY
