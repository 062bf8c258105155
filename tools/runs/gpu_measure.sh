#!/bin/bash
# Round measurement bundle: bench line, rocprofv3 kernel stats of the same command, PMC traffic of the dominant kernel.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1
cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof.log 2>&1
cd $R; cp /tmp/prof/r01_kernel_stats.csv gpurun_out/kernel_stats.csv
# PMC (separate passes, counters only): HBM bytes per launch of the dominant kernel on its headline shape
bash tools/runs/gpu_pmc.sh "conv3d 128->128,conv3d 256->256"
tail -2 gpurun_out/bench.log | cut -c1-600; head -12 gpurun_out/kernel_stats.csv | cut -c1-120
