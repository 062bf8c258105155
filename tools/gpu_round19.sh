#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "attention" 2>&1 | tail -8 > gpurun_out/pytest_gpu.log
rm -f gpurun_out/ablate.log
for Q in 2 1 2 1; do echo "== ATTN_QB=$Q" >> gpurun_out/ablate.log; DOVE_ATTN_QB=$Q timeout 300 python tools/microbench.py --only "attention" 2>&1 | grep -v amdgpu.ids | grep attention >> gpurun_out/ablate.log; done
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/ablate.log
