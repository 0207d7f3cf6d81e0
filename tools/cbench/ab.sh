for lib in hallo_amd tools/cbench/alt; do
echo "== $lib"
for sh in "65536 1280 320 geglu ln" "65536 960 320 ln" "73728 960 320 ln" "65536 320 320"; do
LD_LIBRARY_PATH=$lib tools/cbench/cbench gemm $sh | grep "^gemm"
done; done
