#!/bin/bash
# attention: running maximum (14) vs constant shift from a score bound, compile-time (44) / run-time selected (54) - profiles/r03_attn_fixed_bound.log
mkdir -p gpurun_out
timeout -k 10 200 python tools/archive/attn_nw.py 14 44 54 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_attn_fixed_bound.log
timeout -k 10 200 python tools/archive/attn_nw.py 54 44 14 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_attn_fixed_bound.log
