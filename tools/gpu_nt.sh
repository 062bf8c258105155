#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python tools/e2e_env_ab.py 32 0 5 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r03_gemm8p_nt.log
timeout -k 10 400 python tools/e2e_env_ab.py 16 32 5 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a gpurun_out/r03_gemm8p_nt.log
