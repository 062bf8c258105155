#!/bin/bash
# round-2 check bundle: new parity tests first (with their printed numbers), then the rest of the GPU suite, smoke, bench
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc > gpurun_out/host.log; free -g | head -2 >> gpurun_out/host.log
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_e2e_gpu.py -m gpu -x -q -s -p no:cacheprovider > gpurun_out/pytest_parity.log 2>&1
timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/pytest_ops.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1
grep -v "^$" gpurun_out/pytest_parity.log | grep "^\[\|passed\|failed\|Error\|assert" | cut -c1-700 | tail -40
cat gpurun_out/pytest_ops.log gpurun_out/smoke.log gpurun_out/host.log; tail -1 gpurun_out/bench.log | cut -c1-2500
