#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
timeout 600 python tools/microbench.py --only "attention,linear,qkv_post" > gpurun_out/microbench.log 2>&1
timeout 1200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench.log 2>&1
cat gpurun_out/pytest_gpu.log | tail -12; cat gpurun_out/microbench.log | tail -8; tail -2 gpurun_out/bench.log
