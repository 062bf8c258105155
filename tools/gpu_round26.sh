#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ablate.log
DOVE_CONV_HALO4X=1 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -8 >> gpurun_out/ablate.log
DOVE_HALO4X_GRID=0 DOVE_CONV_HALO4X=1 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -3 >> gpurun_out/ablate.log
DOVE_HALO4X_GRID=7 DOVE_CONV_HALO4X=1 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -3 >> gpurun_out/ablate.log
for X in 256 0 256 0; do echo "== HALO4X grid=$X" >> gpurun_out/ablate.log; DOVE_HALO4X_GRID=$X DOVE_CONV_HALO4X=1 timeout 300 python tools/microbench.py --only "conv3d 128->128,conv3d 256->256,conv3d 512,conv3d 256->128,conv2d up" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
echo "== HALO8" >> gpurun_out/ablate.log; timeout 300 python tools/microbench.py --only "conv3d 128->128,conv3d 256->256,conv3d 512,conv3d 256->128,conv2d up" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log
cat gpurun_out/ablate.log
