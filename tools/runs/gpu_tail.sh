#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 200 python tools/archive/gemm_tail_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_gemm_tail_probe.log
