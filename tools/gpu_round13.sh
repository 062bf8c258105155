#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/pytest_gpu.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1
cd $R; find /tmp/prof -name "*kernel_stats*" -exec cp {} gpurun_out/kernel_stats.csv \; ; ls /tmp/prof/* | head
tail -5 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/bench.log; head -25 gpurun_out/kernel_stats.csv | cut -c1-150
