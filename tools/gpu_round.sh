#!/bin/bash
# One GPU-box session: parity tests (all, no -x), smoke, micro-benchmarks.  Logs land in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" >> gpurun_out/smoke.log
timeout 900 python tools/microbench.py > gpurun_out/microbench.log 2>&1; echo "microbench exit: $?" >> gpurun_out/microbench.log
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -30 gpurun_out/microbench.log
