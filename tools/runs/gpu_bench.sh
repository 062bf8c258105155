#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench exit: $?" >> gpurun_out/bench.log
tail -4 gpurun_out/bench.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo "prof exit: $?" >> $GRAFT_REPO_ROOT/gpurun_out/prof.log
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*stats*" | head; find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete; ls -la gpurun_out/prof/* | head -20
