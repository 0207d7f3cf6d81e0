#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
( timeout 300 tools/cbench/cbench ff 65536 v=1,9; timeout 100 tools/cbench/cbench ff 389 v=9 ) > $O/r3e_cbench_ff.log 2>&1
grep "^ff\|stamps" $O/r3e_cbench_ff.log
