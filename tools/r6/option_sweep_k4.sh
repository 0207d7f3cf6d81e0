#!/bin/bash
# round 6: single-option sweep in the default execution (four clips per evaluation), one box, baseline repeated
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6_o
O=gpurun_out/r6_o
B="--no-cpu-baseline --no-configs2 --no-fp16-leg --no-serial-leg --no-profile --steps 12"
run() { tag=$1; shift; timeout 300 python bench.py $B "$@" > $O/$tag.json 2> $O/$tag.err; echo "$tag $(grep -o '"value": [0-9.]*' $O/$tag.json | head -1)"; }
run base_1
run v3min128 --set-option v3_min_tiles=128
run v3min256 --set-option v3_min_tiles=256
run stage320 --set-option gemm_stage_min_tiles=320
run stage1280 --set-option gemm_stage_min_tiles=1280
run attn40_2 --set-option attn40=2
run attn40_3 --set-option attn40=3
run attn40_5 --set-option attn40=5
run base_2
run tok1 --set-option tok_attn=1
run xattn1 --set-option xattn_tiled=1
run rowparts2 --set-option row_parts=2
run producer0 --set-option producer_stats=0
run gemm4_2 --set-option gemm4=2
run gemm4nk20 --set-option gemm4_min_nk=20
run rs1 --set-option gemm_rs=1
run base_3
