#!/bin/bash
# gemm4.hip (exact-fit / stream-K kernel) against the kernels it replaces: full-matrix comparison + timing, per shape
C=tools/cbench/cbench
run() { for g in 2 0; do $C gemm "$@" g4=$g | grep -E "^gemm|full matrix"; done; }
run 4096 1280 1280 res
run 4096 1280 1280
run 4608 1280 1280
run 4096 1280 5120 res
run 4096 3840 1280 ln
run 4608 3840 1280 ln
run 16384 640 640 res
run 16384 640 2560 res
run 18432 1920 640 ln
run 16384 1920 640 ln
run 4096 1920 640 ln
run 4096 640 2560 res
run 1024 1280 1280 res
run 1024 1280 5120 res
run 1152 3840 1280 ln
run 65536 320 1280 res
run 16384 960 320
run 1000 1288 1280 res
run 5000 648 704 ln
run 4096 1280 1280 res f16
run 4608 3840 1280 ln f16
