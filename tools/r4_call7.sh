#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c7; mkdir -p $O
cd $R
(for sh in "4096 1280 1280 res" "4096 1280 1280" "4096 1280 5120 res"; do tools/cbench/cbench gemm $sh g4=2 stamps nocheck | grep -E "^gemm|stamps"; done) > $O/stamps.txt 2>&1
cat $O/stamps.txt
