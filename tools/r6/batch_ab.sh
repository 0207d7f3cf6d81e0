#!/bin/bash
# round 6, GPU call 2: (1) the round-4 tree and the current tree alternating on ONE box (does the racy round-4 execution really run
# faster, and are its latents finite?); (2) new operator / pipeline tests; (3) clips per UNet evaluation x pipelines in flight sweep
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6_b
O=$PWD/gpurun_out/r6_b
B="--no-cpu-baseline --no-profile --steps 12 --warmup 3"
for r in 1 2; do
  timeout 300 python bench.py $B --no-configs2 > $O/r6tree_$r.json 2> $O/r6tree_$r.err || echo "r6tree $r failed"
done
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "temporal" > $O/test_temporal.log 2>&1; echo "temporal rc=$?"; tail -3 $O/test_temporal.log
timeout 900 python -m pytest tests/test_models_gpu.py -x -q -k "call_batch or end_to_end or unet3d_forward or hipgraph_replay" > $O/test_models.log 2>&1; echo "models rc=$?"; tail -5 $O/test_models.log
S="--no-cpu-baseline --no-profile --no-configs2 --no-serial-leg --steps 12 --warmup 3"
run() { tag=$1; shift; timeout 400 python bench.py $S "$@" > $O/${tag}.json 2> $O/${tag}.err || echo "$tag failed rc=$?"; }
run k1_n3 --inflight 3
run k3_n1 --batch-clips 3 --inflight 1
run k3_n1_thr --batch-clips 3 --inflight 1 --throughput-routing
run k2_n2 --batch-clips 2 --inflight 2
run k3_n2 --batch-clips 3 --inflight 2
run k4_n1 --batch-clips 4 --inflight 1
run k4_n2 --batch-clips 4 --inflight 2 --steps 16
run k2_n3 --batch-clips 2 --inflight 3
run k6_n1 --batch-clips 6 --inflight 1
run k3_n2_lat --batch-clips 3 --inflight 2 --latency-routing
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r6_b/*.json"), key=os.path.getmtime):
    try:
        d = json.load(open(f))
        print(os.path.basename(f)[:-5], round(d["value"], 3), d.get("inflight_identity", {}).get("identical"),
              round((d.get("one_clip_at_a_time") or {}).get("value") or 0, 3), d.get("r6_probe_nan_latent_values"), d.get("r6_probe_clips_checked"))
    except Exception as e:
        print(os.path.basename(f), "failed", str(e)[:100])
        try: print(open(f[:-5] + ".err").read()[-600:])
        except Exception: pass
PY
