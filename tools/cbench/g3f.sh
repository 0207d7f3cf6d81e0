for lib in hallo_amd tools/cbench/alt; do
echo "== $lib"
for sh in "8192 8192 8192" "16384 1280 11520" "65536 320 2880" "16384 640 5760" "4096 5120 1280 geglu ln" "65536 320 1280 res"; do
LD_LIBRARY_PATH=$lib tools/cbench/cbench gemm $sh variant=4 nocheck | grep "^gemm"
done; done
