#!/bin/bash
# gemm8p output stores: default policy vs nontemporal - back-to-back launches, then the whole operator (profiles/r03_gemm8p_nt.log)
mkdir -p gpurun_out
timeout -k 10 200 python tools/archive/gemm8p_nt.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_gemm8p_nt.log
timeout -k 10 400 python tools/e2e_env_ab.py DOVE_IGEMM_ABLATE 16 0 4 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r03_gemm8p_nt.log
