#!/bin/bash
# full GPU suite + smoke + default bench line at HEAD (no profiler passes)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|Error" | tail -4 > gpurun_out/r03_pytest_gpu_head.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r03_pytest_gpu_head.log
if [ "$1" != "tests-only" ]; then timeout 900 python bench.py 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r03_bench_head.log; fi
cat gpurun_out/r03_pytest_gpu_head.log
