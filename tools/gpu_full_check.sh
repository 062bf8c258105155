#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log; tail -1 gpurun_out/bench.log | cut -c1-1500
