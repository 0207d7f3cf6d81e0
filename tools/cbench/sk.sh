for sh in "4096 1280 5120 res" "4608 1280 5120 res" "16384 640 2560 res" "18432 640 2560 res" "4096 1280 1280 res" "4096 640 2560 res" "1024 1280 5120 res" "1024 1280 1280 res"; do
for v in -1 4 5; do
if [ $v = -1 ]; then tools/cbench/cbench gemm $sh | grep "^gemm"; else tools/cbench/cbench gemm $sh variant=$v | grep "^gemm"; fi
done; done
