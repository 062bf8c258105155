#!/bin/bash
# round 3: attention two-wave-group experiment (tools/exp/attn2g_exp.hip) vs the product kernel, within one run
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 600 python tools/archive/attn_ab.py 40,41,42,43,44 2>&1 | grep -v amdgpu | tee gpurun_out/r03_attn2g.log | tail -30
