#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ablate.log
for C in 1 2 3; do DOVE_HALO4X_CFG=$C DOVE_CONV_HALO4X=1 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -2 >> gpurun_out/ablate.log; done
for X in 0 1 2 3 0 1 2 3; do echo "== HALO4X cfg=$X" >> gpurun_out/ablate.log; DOVE_HALO4X_CFG=$X DOVE_CONV_HALO4X=1 timeout 300 python tools/microbench.py --only "conv3d 128->128,conv3d 256->256,conv3d 512,conv3d 256->128,conv2d up" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
echo "== HALO8" >> gpurun_out/ablate.log; timeout 300 python tools/microbench.py --only "conv3d 128->128,conv3d 256->256,conv3d 512,conv3d 256->128,conv2d up" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log
cat gpurun_out/ablate.log
