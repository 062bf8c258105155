#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv or linear" 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
rm -f gpurun_out/ablate.log
for G in 1 0 1 0; do echo "== GEMM8=$G" >> gpurun_out/ablate.log; DOVE_GEMM8=$G timeout 300 python tools/microbench.py --only "linear" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
cat gpurun_out/pytest_gpu.log | tail -20; cat gpurun_out/ablate.log
