#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/r03_qkv_post.log
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_graph_gpu.py tests/test_dist_gpu.py -x -q -m gpu -k "qkv_post or attention or dit or sharded or dist or ulysses" 2>&1 | tail -4 > $L
timeout -k 10 300 python tools/microbench.py --only qkv_post 2>&1 | grep -v amdgpu.ids | grep qkv_post >> $L
cat $L
