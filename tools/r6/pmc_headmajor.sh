#!/bin/bash
# the PMC traffic passes and the cbench timing of the dominant kernel with head-major K / V (the first block of tools/round_end_gpu.sh)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp CBENCH_KV_HM=1
for c in 0 1 3 4; do CBENCH_CASE=$c PMC_GROUPS="fetch write hit" tools/cbench/pmc.sh a40_c$c attn-time 1; done
python tools/pmc_traffic_cbench.py gpurun_out > gpurun_out/r6_pmc_traffic.json
timeout 200 tools/cbench/cbench attn-time 1 > gpurun_out/r6_attn_time.txt 2>&1
unset CBENCH_KV_HM
timeout 200 tools/cbench/cbench attn-time 1 > gpurun_out/r6_attn_time_rowmajor.txt 2>&1
grep "bf16" gpurun_out/r6_attn_time.txt gpurun_out/r6_attn_time_rowmajor.txt | cut -c1-200
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_pmc_traffic.json"))
for e in d.get("attn40_kernel", []):
    print(e["launch"][:60], e["fetch_over_algorithmic"], e["write_over_algorithmic"], e.get("l2_hit_rate"))
PY
