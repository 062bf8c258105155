#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/timing.log
for A in 0 8 40 1 2 3; do echo "== DEBUG=$A" >> gpurun_out/timing.log; DOVE_IGEMM_ABLATE=$A timeout 200 python tools/halo8_timing.py 128 128 2>&1 | grep -v amdgpu.ids | grep -v per-step >> gpurun_out/timing.log; done
rm -f gpurun_out/ablate.log
for A in 0 8 40; do echo "== DEBUG=$A" >> gpurun_out/ablate.log; DOVE_IGEMM_ABLATE=$A timeout 300 python tools/microbench.py --only "conv3d 128->128,conv3d 256->256" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
cat gpurun_out/timing.log; cat gpurun_out/ablate.log
