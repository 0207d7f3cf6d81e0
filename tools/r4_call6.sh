#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c6; mkdir -p $O
cd $R
(for sh in "4096 1280 1280 res" "4096 1280 5120 res" "4096 3840 1280 ln" "4608 1280 1280"; do tools/cbench/cbench gemm $sh g4=2 stamps nocheck | grep -E "^gemm|stamps|full"; done) > $O/stamps.txt 2>&1
timeout 600 bash tools/cbench/g4.sh > $O/g4.txt 2>&1
cat $O/stamps.txt
cat $O/g4.txt | grep -v "^  full" | awk '{print $2,$3,$4,$5,$6, $11, $14,$15,$16,$17}' 
grep "full matrix" $O/g4.txt | awk '{print $6,$7,$8,$11,$15}' | sort | uniq -c | sort -rn | head -30
