#!/bin/bash
# round 4, first GPU call: (1) which synthetic weight tensors does this host draw differently (default vs avx2 ATen dispatch);
# (2) baseline bench of the round-3 binary; (3) PMC passes over the small-M GEMM shapes
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c1; mkdir -p $O
cd $R
lscpu | grep -i "model name\|^CPU(s)" > $O/host.txt
(python tools/weights_fingerprint.py $O/bits_default.json tests/golden/weights_per_tensor_bits.json > $O/fp_default.log 2>&1 &
 ATEN_CPU_CAPABILITY=avx2 python tools/weights_fingerprint.py $O/bits_avx2.json tests/golden/weights_per_tensor_bits.json > $O/fp_avx2.log 2>&1 &
 wait) &
FP=$!
python bench.py --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err
wait $FP
tools/cbench/pmc.sh g_4096_1280_1280 gemm 4096 1280 1280 res nocheck
tools/cbench/pmc.sh g_4096_1280_5120 gemm 4096 1280 5120 res nocheck
tools/cbench/pmc.sh g_1024_1280_5120 gemm 1024 1280 5120 res nocheck
tools/cbench/pmc.sh g_4096_1920_640_ln gemm 4096 1920 640 ln nocheck
bash tools/cbench/sk.sh > $O/sk.txt 2>&1
for d in gpurun_out/pmc_g_*; do cp $d/summary.txt $O/$(basename $d).txt; done
tail -c 600 $O/bench.json
