#!/bin/bash
# Round-5 call B: the frame-pair (tdup) conv tests, the pipelined attention experiment (correctness + timing, real and zero operands), and the
# whole-operator A/B of the frame-pair sums (vae.weight_sums on: with w_pair; off: per-tap arithmetic everywhere).
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider -k "tdup or test_conv or w_first" 2>&1 | grep -a "weight sums\|passed\|failed\|Error\|error\|assert" | cut -c1-500 > gpurun_out/r05_b_tests.log
timeout 300 python tools/attn2p_ab.py 2>&1 | grep -v amdgpu > gpurun_out/r05_attn2p.log
timeout 200 python tools/attn2p_ab.py zeros 2>&1 | grep -v amdgpu | tail -2 >> gpurun_out/r05_attn2p.log
timeout 400 python tools/tdup_ab.py 2>&1 | grep -v amdgpu > gpurun_out/r05_tdup_ab.log
cat gpurun_out/r05_b_tests.log gpurun_out/r05_attn2p.log gpurun_out/r05_tdup_ab.log
