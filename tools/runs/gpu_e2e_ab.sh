#!/bin/bash
# whole-operator A/B of one timing-library switch on the headline clip: gpu_e2e_ab.sh <VAR> <A> <B> [rounds]
#   DOVE_GEMM8P 0 1        gemm4x vs gemm8p                         (profiles/r03_e2e_gemm8p_ab.log)
#   DOVE_ATTN_BOUND 0 1    running maximum vs score bound           (profiles/r03_attn_bound.log)
#   DOVE_IGEMM_ABLATE 64 0 conv epilogue without / with the early slice write (profiles/r03_halo4x_epilogue.log)
mkdir -p gpurun_out
timeout -k 10 400 python tools/e2e_env_ab.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/e2e_ab_$1.log
