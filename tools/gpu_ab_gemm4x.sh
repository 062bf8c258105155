#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab.log
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "linear or conv" 2>&1 | tail -8 >> gpurun_out/ab.log
for X in 1 0 1 0; do echo "== GEMM4X=$X" >> gpurun_out/ab.log; DOVE_GEMM4X=$X timeout 300 python tools/microbench.py --only "linear" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab.log; done
cat gpurun_out/ab.log
