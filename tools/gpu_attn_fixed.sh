#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 200 python tools/attn_nw.py 54 44 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_attn_exp_legacy.log
