#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ablate.log
for X in 0 128 0 128; do echo "== HALO4X persistent ablate=$X" >> gpurun_out/ablate.log; DOVE_IGEMM_ABLATE=$X DOVE_CONV_HALO4X=1 timeout 300 python tools/microbench.py --only "conv3d 128->128,conv3d 256->256,conv3d 512,conv3d 256->128,conv2d up" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
cat gpurun_out/ablate.log
