#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/attn2p_ab.py real 0,2,8,64,66,72,128 2>&1 | grep -v amdgpu > gpurun_out/r05_attn2p_e.log
timeout 200 python tools/attn2p_ab.py zeros 0,2,8,64,66,72,128 2>&1 | grep -v amdgpu >> gpurun_out/r05_attn2p_e.log
cat gpurun_out/r05_attn2p_e.log
