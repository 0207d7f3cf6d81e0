#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c5; mkdir -p $O
cd $R
timeout 600 bash tools/cbench/g4.sh > $O/g4.txt 2>&1
cat $O/g4.txt | grep -v "^  full" | awk '{print $2,$3,$4,$5,$6, $11, $14,$15,$16,$17}' 
grep "full matrix" $O/g4.txt | awk '{print $6,$7,$8,$11,$15}' | sort | uniq -c | sort -rn | head -30
