#!/bin/bash
# round 5, GPU call: host CPU vs AQL queue size; prefetch A/B of the token / face kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
timeout 300 python tools/r5_pf_ab.py > $O/r5_pf_ab.log 2>&1; echo "pf rc=$?"; cat $O/r5_pf_ab.log | cut -c1-400
B="--no-cpu-baseline --no-profile --no-serial-leg --no-configs2 --steps 9 --warmup 3"
for q in default 65536 131072; do
  if [ $q = default ]; then unset ROC_AQL_QUEUE_SIZE; else export ROC_AQL_QUEUE_SIZE=$q; fi
  timeout 300 python bench.py $B > $O/r5_bench_aql_$q.json 2> $O/r5_bench_aql_$q.err
  echo "aql $q rc=$?"; python - <<PY
import json
d=json.load(open("gpurun_out/r5_bench_aql_$q.json")); c=d["config"]
print(round(d["value"],2), {k:c[k] for k in c if k.startswith("host_c") or k.startswith("host_w")})
PY
done
export ROC_AQL_QUEUE_SIZE=131072
export HIP_FORCE_DEV_KERNARG=1
timeout 300 python bench.py $B > $O/r5_bench_aql_devkernarg.json 2> $O/r5_bench_aql_devkernarg.err
python - <<PY
import json
d=json.load(open("gpurun_out/r5_bench_aql_devkernarg.json")); c=d["config"]
print("devkernarg", round(d["value"],2), {k:c[k] for k in c if k.startswith("host_c") or k.startswith("host_w")})
PY
