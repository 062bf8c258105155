#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -12 > gpurun_out/pytest_gpu.log
rm -f gpurun_out/ablate.log
for H8 in 1 0; do echo "== HALO8=$H8" >> gpurun_out/ablate.log; DOVE_CONV_HALO8=$H8 timeout 300 python tools/microbench.py --only "conv2d up" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
tail -8 gpurun_out/pytest_gpu.log; cat gpurun_out/ablate.log
