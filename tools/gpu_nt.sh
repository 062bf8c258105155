#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python tools/e2e_env_ab.py DOVE_GEMM8P 0 1 5 2>&1 | grep -v amdgpu.ids | tail -1 | tee gpurun_out/r03_e2e_gemm8p_ab.log
