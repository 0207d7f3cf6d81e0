#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6_p
O=gpurun_out/r6_p
timeout 900 python -m pytest tests/test_full_size_gpu.py -x -q -k "unet3d_forward or call_batch or (trajectory and pipeline25)" > $O/t_full.log 2>&1; echo "full rc=$?"; tail -2 $O/t_full.log
timeout 600 python -m pytest tests/test_models_gpu.py -x -q -k "unet3d_forward or end_to_end" > $O/t_models.log 2>&1; echo "models rc=$?"; tail -2 $O/t_models.log
B="--no-cpu-baseline --no-configs2 --no-fp16-leg --no-serial-leg --no-profile --steps 12"
for r in 1 2 3; do
timeout 400 python bench.py $B > $O/kpad64_$r.json 2> $O/kpad64_$r.err; grep -o '"value": [0-9.]*' $O/kpad64_$r.json | head -1 | sed 's/^/kpad64 /'
timeout 400 python bench.py $B --audio-kpad8 > $O/kpad8_$r.json 2> $O/kpad8_$r.err; grep -o '"value": [0-9.]*' $O/kpad8_$r.json | head -1 | sed 's/^/kpad8 /'
done
