#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 150 python tools/archive/gemm8p_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_gemm8p_ab.log
