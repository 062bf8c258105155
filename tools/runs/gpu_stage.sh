#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 200 python tools/archive/stage_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_stage_ab.log
