#!/bin/bash
# full GPU suite + smoke + driver-like bench line at HEAD (no profiler passes)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -x -q -s -p no:cacheprovider 2>&1 | grep -a "weight sums\]\|\[heavy x4\]\|\[bench single\|\[tol\]\|passed\|failed\|Error\|error\|assert" | cut -c1-600 > gpurun_out/r05_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 > gpurun_out/r05_smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r05_bench_full.log
cat gpurun_out/r05_pytest_gpu.log gpurun_out/r05_smoke.log; tail -1 gpurun_out/r05_bench_full.log | cut -c1-400
