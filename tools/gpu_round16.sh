#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
rm -f gpurun_out/timing.log gpurun_out/ablate.log
for A in 0 64; do echo "== DEBUG=$A" >> gpurun_out/timing.log; DOVE_IGEMM_ABLATE=$A timeout 200 python tools/halo8_timing.py 128 128 2>&1 | grep -v amdgpu.ids >> gpurun_out/timing.log; done
for A in 0 64 0 64; do echo "== DEBUG=$A" >> gpurun_out/ablate.log; DOVE_IGEMM_ABLATE=$A timeout 300 python tools/microbench.py --only "conv3d 128->128,conv3d 256->256,conv3d 512,conv3d 256->128" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/timing.log; cat gpurun_out/ablate.log
