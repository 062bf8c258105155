#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab.log
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "attention or qkv" 2>&1 | tail -3 >> gpurun_out/ab.log
for X in 1 1; do timeout 300 python tools/microbench.py --only "attention" 2>&1 | grep "attention" >> gpurun_out/ab.log; done
cat gpurun_out/ab.log
