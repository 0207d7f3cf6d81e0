# LN-fused long-K shapes: auto rule (now may take gemm3) vs forced 128x128 (variant 3) vs forced gemm3 TM=2 (4) / TM=1 (5)
for sh in "4096 5120 1280 geglu ln" "4608 5120 1280 geglu ln" "4096 3840 1280 ln" "4608 3840 1280 ln" "16384 2560 640 geglu ln" "18432 2560 640 geglu ln" "16384 1920 640 ln" "18432 1920 640 ln" "1024 5120 1280 geglu ln"; do
for v in -1 3 4 5; do
if [ $v = -1 ]; then tools/cbench/cbench gemm $sh | grep "^gemm"; else tools/cbench/cbench gemm $sh variant=$v | grep "^gemm"; fi
done; done
