for d in 0 1 2 3 4 6 8 12 14 15; do tools/cbench/cbench gemm 65536 960 320 ln nocheck rsdbg=$d | grep "^gemm"; done
for d in 0 1 2 3 4 6 8 12 14 15; do tools/cbench/cbench gemm 65536 1280 320 geglu ln nocheck rsdbg=$d | grep "^gemm"; done
tools/cbench/cbench gemm 65536 960 320 ln nocheck rs=0 | grep "^gemm"
tools/cbench/cbench gemm 65536 1280 320 geglu ln nocheck rs=0 | grep "^gemm"
tools/cbench/cbench gemm 65536 1920 640 ln nocheck | grep "^gemm"
tools/cbench/cbench gemm 65536 1920 640 ln nocheck rs=0| grep "^gemm"
