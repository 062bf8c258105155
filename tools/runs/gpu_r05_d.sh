#!/bin/bash
# Round-5 call D: the pipelined attention kernel in the product - its operator tests, the DiT parity tests that run through it, the experiment
# twin's A/B (correctness on ragged lengths + timing by parts) and the whole-operator A/B of the switch (timing library: DOVE_ATTN_PIPE 0 1).
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_parity_gpu.py tests/test_graph_gpu.py tests/test_dist_gpu.py -x -q -m gpu -p no:cacheprovider \
  -k "attention or qkv_post or dit_42 or mixed_softmax or north_star or one_clip_sharded or sharded_on_hip or test_dit or sr_clip" 2>&1 | grep -a "passed\|failed\|Error\|error\|assert" | cut -c1-400 > gpurun_out/r05_d_tests.log
timeout 300 python tools/attn2p_ab.py real 0,2,8,72 2>&1 | grep -v amdgpu > gpurun_out/r05_attn2p_f.log
timeout 200 python tools/attn2p_ab.py zeros 0,8,72 2>&1 | grep -v amdgpu | tail -5 >> gpurun_out/r05_attn2p_f.log
timeout 400 python tools/e2e_env_ab.py DOVE_ATTN_PIPE 0 1 4 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r05_e2e_attn_pipe.log
cat gpurun_out/r05_d_tests.log; grep -v "^N =" gpurun_out/r05_attn2p_f.log; cat gpurun_out/r05_e2e_attn_pipe.log
