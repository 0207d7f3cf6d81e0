#!/bin/bash
# round 6, GPU call 13: clips per evaluation x pipelines in flight x routing, finer, two rounds, one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6_l
O=$PWD/gpurun_out/r6_l
S="--no-cpu-baseline --no-profile --no-configs2 --no-serial-leg --no-fp16-leg --warmup 4"
run() { tag=$1; shift; timeout 400 python bench.py $S "$@" > $O/${tag}.json 2> $O/${tag}.err || echo "$tag failed rc=$?"; }
for r in 1 2; do
  run k1_n3_$r --inflight 3 --steps 12
  run k4_n1_$r --batch-clips 4 --inflight 1 --steps 12
  run k2_n1_$r --batch-clips 2 --inflight 1 --steps 12
  run k3_n1_$r --batch-clips 3 --inflight 1 --steps 12
  run k5_n1_$r --batch-clips 5 --inflight 1 --steps 15
  run k8_n1_$r --batch-clips 8 --inflight 1 --steps 16
  run k2_n2_lat_$r --batch-clips 2 --inflight 2 --latency-routing --steps 12
  run k4_n1_gemm4off_$r --batch-clips 4 --inflight 1 --steps 12 --set-option gemm4=0
  run k4_n1_rs0_$r --batch-clips 4 --inflight 1 --steps 12 --set-option gemm_rs=0
  run k4_n1_ff1_$r --batch-clips 4 --inflight 1 --steps 12 --set-option ff_fused=1
  run k4_n1_gn0_$r --batch-clips 4 --inflight 1 --steps 12 --set-option gn_fused=0
  run k4_n1_sk4_$r --batch-clips 4 --inflight 1 --steps 12 --set-option split_k_max=4
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r6_l/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f)[:-5], round(d["value"], 3), d.get("inflight_identity", {}).get("identical"))
    except Exception as e:
        print(os.path.basename(f), "failed", str(e)[:100])
        try: print(open(f[:-5] + ".err").read()[-400:])
        except Exception: pass
PY
