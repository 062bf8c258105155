#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/r03_attn_bound.log
timeout -k 10 1500 python -m pytest tests/test_ops_gpu.py tests/test_graph_gpu.py tests/test_dist_gpu.py tests/test_parity_gpu.py tests/test_abi.py -x -q -m gpu -k "attention or qkv_post or dit or graph or sharded or dist or north_star or stagewise or abi or sr_clip" 2>&1 | tail -6 > $L
timeout -k 10 400 python tools/e2e_env_ab.py DOVE_ATTN_BOUND 0 1 5 2>&1 | grep -v amdgpu.ids | tail -1 >> $L
cat $L
