#!/bin/bash
# Round-5 call E: the one-block-per-wave form of the pipelined attention kernel for the last partial round of workgroups: attention operator
# tests (small / ragged lengths take the NB = 1 form alone or beside NB = 2), the tool's "product" arm (library: NB = 2 rounds + NB = 1 tail)
# against its experiment arm (NB = 2 for every item), DiT / sharded bit-identity tests.
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/attn2p_ab.py real 0 2>&1 | grep -v amdgpu > gpurun_out/r05_attn_tail.log
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_parity_gpu.py tests/test_graph_gpu.py tests/test_dist_gpu.py -x -q -m gpu -p no:cacheprovider \
  -k "attention or qkv_post or dit_42 or mixed_softmax or one_clip_sharded or sharded_on_hip or test_dit or sr_clip or dit_sharded_42" 2>&1 | grep -a "passed\|failed\|Error\|error\|assert" | cut -c1-400 >> gpurun_out/r05_attn_tail.log
cat gpurun_out/r05_attn_tail.log
