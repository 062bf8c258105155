#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
rm -f gpurun_out/ablate.log
for H8 in 1 0 1 0; do echo "== HALO8=$H8" >> gpurun_out/ablate.log; DOVE_CONV_HALO8=$H8 timeout 300 python tools/microbench.py --only "conv3d 128->128,conv3d 256->256,conv3d 512,conv3d 256->128" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
cat gpurun_out/pytest_gpu.log | tail -20; cat gpurun_out/ablate.log
