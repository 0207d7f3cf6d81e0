#!/bin/bash
# head-major K / V of the 64 x 64-level self-attentions (hallo_gemm kv_out + hallo_attention head strides): operator tests, the
# pipeline tests that run through it, and the end-to-end A/B on one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6_headmajor; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "kv_split or head_major or test_attention or call_batch or full_unet3d_forward or pipeline_end_to_end or full_pipeline_trajectory" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
B="python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-configs2 --no-fp16-leg --no-serial-leg"
for r in 1 2; do
  timeout 300 $B > $O/bench_hm_$r.log 2>&1; echo "head-major $r $(grep -o '"value": [0-9.]*' $O/bench_hm_$r.log | head -1)"
  timeout 300 $B --no-kv-head-major > $O/bench_rowmajor_$r.log 2>&1; echo "row-major $r $(grep -o '"value": [0-9.]*' $O/bench_rowmajor_$r.log | head -1)"
done
python - <<'PY'
import json
for f in ("hm_1", "rowmajor_1"):
    try:
        d = json.loads([l for l in open("gpurun_out/r6_headmajor/bench_%s.log" % f) if l.startswith("{")][-1])
        print(f, d["value"], d.get("inflight_identity", {}).get("identical"), d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"),
              {k: v["ms"] for k, v in d.get("kernels", {}).items() if k in ("gemm", "attention")})
    except Exception as e:
        print(f, "no line", e)
PY
