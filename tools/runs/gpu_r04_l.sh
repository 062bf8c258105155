#!/bin/bash
# round 4, call L: rocprofv3 kernel stats of the tiled VAE pass at the HEAD of the round (batched, edge classes on a second stream)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tiled
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tiled -o tiled -- python $R/tools/tiled_bench.py --reps 1 --mode tiled > $R/gpurun_out/r04_tiled_prof2.log 2>&1
cp /tmp/prof_tiled/tiled_kernel_stats.csv $R/gpurun_out/r04_tiled_final_kernel_stats.csv
head -8 $R/gpurun_out/r04_tiled_final_kernel_stats.csv | cut -c1-140; tail -1 $R/gpurun_out/r04_tiled_prof2.log | cut -c1-300
