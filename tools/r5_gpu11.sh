#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
B="--no-cpu-baseline --no-profile --no-serial-leg --no-configs2 --warmup 4"
for n in 3 4 2 5 3 4; do
  timeout 300 python bench.py $B --inflight $n --steps 12 > $O/r5_inflight_${n}_$RANDOM.json 2> $O/r5_inflight.err
done
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r5_inflight_*.json"), key=os.path.getmtime):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["config"]["clips_in_flight_per_gpu"], round(d["value"],3), d["inflight_identity"]["identical"], d["config"]["host_cores_busy_per_rank"])
    except Exception as e: print(f, "failed", e)
PY
