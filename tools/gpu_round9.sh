#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ablate.log
for A in 0 16 32 0 16 32; do echo "== DEBUG=$A" >> gpurun_out/ablate.log; DOVE_IGEMM_ABLATE=$A timeout 300 python tools/microbench.py --only "conv3d 128->128,conv3d 256->256,conv3d 512" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
cat gpurun_out/ablate.log
