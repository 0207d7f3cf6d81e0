#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
rm -f $O/r5_ab_*.json
B="--no-cpu-baseline --no-serial-leg --no-configs2 --no-profile --steps 9 --warmup 3"
for rnd in 1 2; do
for v in "producer_stats=0" "producer_stats=1 row_parts=1" "producer_stats=1 row_parts=2"; do
  so=""; for kv in $v; do so="$so --set-option $kv"; done
  timeout 300 python bench.py $B $so > "$O/r5_ab_$(echo $v | tr ' =' '__')_$rnd.json" 2> $O/r5_ab.err
done; done
timeout 300 python bench.py $B --inflight 1 --set-option producer_stats=0 > $O/r5_ab_single_ps0.json 2>> $O/r5_ab.err
timeout 300 python bench.py $B --inflight 1 --set-option producer_stats=1 > $O/r5_ab_single_ps1.json 2>> $O/r5_ab.err
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r5_ab_*.json")):
    try:
        d = json.load(open(f)); print(os.path.basename(f), round(d["value"], 3), d.get("inflight_identity", {}).get("identical"))
    except Exception as e: print(f, "failed", e)
PY
