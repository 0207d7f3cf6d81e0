#!/bin/bash
# round 5: producer-side LayerNorm statistics -- operator tests, then A/B in the default bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "row_parts or geglu_fused_layernorm or face_xattn_fused or test_gemm_fused_layernorm or gemm_plain or gemm_epilogues" > $O/r5_parts_ops.log 2>&1
echo "ops rc=$?" >> $O/r5_parts_ops.log; tail -15 $O/r5_parts_ops.log
timeout 600 python -m pytest tests/test_models_gpu.py -x -q -k "unet3d_forward or end_to_end or in_flight" > $O/r5_parts_models.log 2>&1
echo "models rc=$?" >> $O/r5_parts_models.log; tail -5 $O/r5_parts_models.log
B="--no-cpu-baseline --no-serial-leg --no-configs2 --steps 9 --warmup 3"
for ps in 1 0 1 0; do
  timeout 300 python bench.py $B --set-option producer_stats=$ps > $O/r5_bench_ps${ps}_$RANDOM.json 2> $O/r5_bench_ps.err
  python - <<PY
import json,glob,os
f=max(glob.glob("gpurun_out/r5_bench_ps${ps}_*.json"), key=os.path.getmtime)
d=json.load(open(f)); k=d.get("kernels",{})
print("producer_stats=$ps", round(d["value"],3), d.get("inflight_identity",{}).get("identical"), {n:(k[n]["ms"],k[n]["launches"]) for n in ("gemm","row_stats","face_xattn") if n in k})
PY
done
