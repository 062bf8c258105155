#!/bin/bash
# round 4, call K: the sub-pixel upsample conv against its direct form at the production shape: timing + PMC passes
mkdir -p gpurun_out
python tools/microbench.py --only "conv2d up" --iters 5 > gpurun_out/r04_up_micro.log 2>&1
rm -f gpurun_out/pmc.log
bash tools/runs/gpu_pmc.sh "conv2d up" > /dev/null 2>&1
cat gpurun_out/r04_up_micro.log | grep conv2d; grep -A40 "conv3x3_halo4x" gpurun_out/pmc_summary.txt | head -60
