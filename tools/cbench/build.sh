#!/bin/sh
# builds tools/cbench/cbench against hallo_amd/libhallo_amd.so (run `python __graft_entry__.py` first)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 cbench.hip -o cbench -L../../hallo_amd -lhallo_amd -Wl,-rpath,'$ORIGIN/../../hallo_amd'
