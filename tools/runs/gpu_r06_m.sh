#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/partial_tile_ab.py 2>&1 | grep -v amdgpu > gpurun_out/r06_m_partial_tile_ab.log
cat gpurun_out/r06_m_partial_tile_ab.log
