#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab.log
DOVE_ATTN_QB=2 timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "attention" 2>&1 | tail -3 >> gpurun_out/ab.log
for X in 2 1 2 1; do echo "== QB=$X" >> gpurun_out/ab.log; DOVE_ATTN_QB=$X timeout 300 python tools/microbench.py --only "attention" 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab.log; done
cat gpurun_out/ab.log
