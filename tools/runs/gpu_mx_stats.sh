#!/bin/bash
# rocprofv3 kernel stats of the BASELINE configs[4] variant (MXFP8 DiT linears + attention) over the bench
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o mx -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --dit-linear mxfp8 --dit-attention mxfp8 > $GRAFT_REPO_ROOT/gpurun_out/bench_mx_prof.log 2>&1
cd $GRAFT_REPO_ROOT; cp /tmp/prof/mx_kernel_stats.csv gpurun_out/kernel_stats_mx.csv
grep -a '"metric"' gpurun_out/bench_mx_prof.log | tail -1 | cut -c1-400; head -12 gpurun_out/kernel_stats_mx.csv | cut -c1-140
