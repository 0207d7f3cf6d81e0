#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4c3; mkdir -p $O
cd $R
timeout 300 python tools/mall_chunk_ab.py > $O/mall.log 2>&1; cp gpurun_out/mall_chunking.json $O/ 2>/dev/null
timeout 600 python -m pytest tests/test_multigpu_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "rccl or reload or hipgraph" > $O/pytest.log 2>&1
tail -3 $O/pytest.log; cat $O/mall.log | tail -12
