#!/bin/bash
# round 4, call A: baseline of the reference's published configuration (--is_vae_st) at 33x720x1280, VAE only, + rocprof kernel stats of the tiled pass
mkdir -p gpurun_out
python tools/tiled_bench.py --reps 3 > gpurun_out/r04_tiled_base.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_tiled -o tiled -- python $GRAFT_REPO_ROOT/tools/tiled_bench.py --reps 1 --mode tiled > $GRAFT_REPO_ROOT/gpurun_out/r04_tiled_prof.log 2>&1
cp $(find /tmp/prof_tiled -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04_tiled_base_kernel_stats.csv
cat $GRAFT_REPO_ROOT/gpurun_out/r04_tiled_base.log | tail -3
