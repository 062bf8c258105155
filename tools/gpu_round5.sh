#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
timeout 600 python tools/microbench.py --only "conv3d" > gpurun_out/microbench.log 2>&1
cat gpurun_out/pytest_gpu.log | tail -30; cat gpurun_out/microbench.log | tail -14
