for z in "" zeroA zeroW "zeroA zeroW"; do
tools/cbench/cbench gemm 65536 1280 320 geglu ln rs=2 nocheck $z | grep "^gemm"
tools/cbench/cbench gemm 65536 1280 320 geglu ln rs=2 nocheck rsdbg=2 $z | grep "^gemm"
done
tools/microbench/mfma_rate 2>&1 | tail -12
