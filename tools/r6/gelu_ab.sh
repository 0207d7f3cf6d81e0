#!/bin/bash
# A/B of the round-6 GELU form (degree-5 exp2 polynomial + packed fp32 pairs) against the A&S 7.1.26 form it replaces:
# tools/cbench/alt/libhallo_amd.so = the library built from the previous commit.  Per launch (cbench) and end to end (bench.py,
# alternating on one box).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r6_gelu_ab; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "geglu or ff320 or big_tile_layernorm or feed_forward" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for lib in tools/cbench/alt hallo_amd tools/cbench/alt hallo_amd; do
  echo "== $lib"
  for a in "ff 262144" "ff 65536 f16" "gemm 65536 2560 640 geglu ln" "gemm 16384 5120 1280 geglu ln" "gemm 65536 1280 320 geglu ln" "gemm 262144 1280 320 geglu ln"; do
    LD_LIBRARY_PATH=$lib timeout 120 tools/cbench/cbench $a 2>&1 | grep "^ff M\|^gemm" | cut -c1-260
  done
done 2>&1 | tee $O/cbench.txt
cp hallo_amd/libhallo_amd.so /tmp/new.so
B="python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-profile --no-configs2 --no-fp16-leg --no-serial-leg"
for r in 1 2; do
  cp /tmp/new.so hallo_amd/libhallo_amd.so; timeout 300 $B > $O/bench_new_$r.log 2>&1; echo "new $r $(grep -o '"value": [0-9.]*' $O/bench_new_$r.log | head -1)"
  cp tools/cbench/alt/libhallo_amd.so hallo_amd/libhallo_amd.so; timeout 300 $B > $O/bench_old_$r.log 2>&1; echo "old $r $(grep -o '"value": [0-9.]*' $O/bench_old_$r.log | head -1)"
done
cp /tmp/new.so hallo_amd/libhallo_amd.so
