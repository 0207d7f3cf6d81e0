cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
B="python bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-configs2 --no-fp16-leg --no-serial-leg"
for cap in 512 256 1024 128; do
  timeout 300 $B --set-option xattn_cap=$cap > gpurun_out/r6_cap_$cap.log 2>&1
  python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r6_cap_$cap.log") if l.startswith("{")][-1])
f = [v for k, v in d["attention_families"].items() if k.startswith("fused face")][0]
print("cap $cap", round(d["value"], 3), "face", f["hbm_frac"], f["ms"])
PY
done
