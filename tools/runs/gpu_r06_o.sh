#!/bin/bash
# round 6, call O: cost side of the GroupNorm + SiLU consumer fusion inside the product conv walk (timing library, kFill)
mkdir -p gpurun_out
timeout 900 python tools/gn_fusion_cost.py 2>&1 | grep -v amdgpu > gpurun_out/r06_o_gn_fusion_cost.log
cat gpurun_out/r06_o_gn_fusion_cost.log
