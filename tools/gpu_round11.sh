#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ablate.log gpurun_out/pmc.log
for A in 0 8 3 4 7; do echo "== ABLATE=$A" >> gpurun_out/ablate.log; DOVE_IGEMM_ABLATE=$A timeout 300 python tools/microbench.py --only "conv3d 128->128,conv3d 256->256" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
bash tools/gpu_pmc.sh "conv3d 128->128"
cat gpurun_out/ablate.log
