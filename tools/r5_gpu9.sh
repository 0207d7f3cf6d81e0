#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
O=gpurun_out
B="--no-cpu-baseline --no-profile --no-serial-leg --no-configs2 --steps 9 --warmup 3"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $O/r5_host_$name.json 2> $O/r5_host_$name.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r5_host_$name.json")); c=d["config"]
    print("$name", round(d["value"],2), d["inflight_identity"]["identical"], "proc", c["host_cpu_ms_per_clip"], "launch", c["host_cpu_launch_thread_ms_per_clip"], "cores", c["host_cores_busy_per_rank"], c["host_cpu_ms_per_clip_by_thread"])
except Exception as e: print("$name failed", e)
PY
}
run base A=1
run cpuwait0 ROC_CPU_WAIT_FOR_SIGNAL=0
run hsaint0 HSA_ENABLE_INTERRUPT=0
run sigpool ROC_SIGNAL_POOL_SIZE=4096
run nodirect AMD_DIRECT_DISPATCH=0
