#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ablate.log
DOVE_CONV_HALO4X=1 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -8 >> gpurun_out/ablate.log
DOVE_HALO4X_GRID=7 DOVE_CONV_HALO4X=1 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -3 >> gpurun_out/ablate.log
for c in "128 128" "256 256"; do DOVE_HALO4X_CFG=9 DOVE_CONV_HALO4X=1 timeout 300 python tools/halo4x_timing.py $c 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/ablate.log; done
for X in 1 0 1; do echo "== HALO4X=$X" >> gpurun_out/ablate.log; DOVE_CONV_HALO4X=$X timeout 300 python tools/microbench.py --only "conv3d 128->128,conv3d 256->256,conv3d 512,conv3d 256->128,conv2d up" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
cat gpurun_out/ablate.log
