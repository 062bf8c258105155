#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
rm -f gpurun_out/ablate.log
for A in 0 8 0 8; do echo "== DEBUG=$A" >> gpurun_out/ablate.log; DOVE_IGEMM_ABLATE=$A timeout 300 python tools/microbench.py --only "conv3d 128->128,conv3d 256->256,conv3d 512" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
cat gpurun_out/pytest_gpu.log | tail -30; cat gpurun_out/ablate.log
