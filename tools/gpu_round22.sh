#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
DOVE_CONV_HALO4X=1 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -12 > gpurun_out/pytest_gpu.log
rm -f gpurun_out/ablate.log
for X in 1 0 1 0; do echo "== HALO4X=$X" >> gpurun_out/ablate.log; DOVE_CONV_HALO4X=$X timeout 300 python tools/microbench.py --only "conv3d 128->128,conv3d 256->256,conv3d 512,conv3d 256->128,conv2d up" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
tail -8 gpurun_out/pytest_gpu.log; cat gpurun_out/ablate.log
