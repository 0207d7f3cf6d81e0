#!/bin/bash
# The alternating A/B runs behind profiles/r4_throughput_options_ab.json, r4_inflight_ab.json, r4_gemm4_e2e_ab.json (one gpurun call
# per block, one box per call):   gpurun --timeout 1500 -- 'bash tools/r4_throughput_ab.sh <block>'
cd "${GRAFT_REPO_ROOT:-$PWD}"
run() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-profile --no-serial-leg "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$*', round(d['value'],3), d['config']['kernel_routing'][:12], d['config']['clips_in_flight_per_gpu'])"; }
case "${1:-routing}" in
  inflight)   for n in 1 2 3 4 6 1 2 3; do run --inflight $n --latency-routing; done ;;
  routing)    for r in 1 2 3 4; do run --latency-routing; run; run --latency-routing --set-option gemm_rs=0; done ;;
  options)    for r in 1 2; do run; run --set-option split_k=0; run --set-option tok_attn=0; run --set-option xattn_tiled=0
                               run --set-option temporal_mfma=1; run --set-option attn40=0; run --inflight 4
                               run --set-option gemm4=1; run --set-option gn_fused=1; run --set-option ff_fused=0; run --set-option gemm_rs=2; done ;;
  tiles)      for r in 1 2; do run; run --gemm-variant 5; run --gemm-variant 4; run --set-option v3_min_tiles=320
                               run --set-option gemm_stage_min_tiles=0; run --set-option gemm_stage_min_tiles=100000; done ;;
  gemm4)      for r in 1 2; do run --inflight 1 --set-option gemm4=0; run --inflight 1; run --inflight 1 --set-option gemm4_min_nk=20; done ;;
esac
