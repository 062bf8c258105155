#!/bin/bash
# round 4, call C: rocprofv3 kernel stats of the batched tiled VAE pass (33x720x1280) + re-run of the bit-identity test
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_e2e_gpu.py -x -q -k "tile_batching" > gpurun_out/r04_c_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r04_c_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tiled -o tiled -- python $R/tools/tiled_bench.py --reps 1 --mode tiled > $R/gpurun_out/r04_tiled_prof.log 2>&1
find /tmp/prof_tiled -name '*.csv' | head
cp $(find /tmp/prof_tiled -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r04_tiled_batched_kernel_stats.csv
tail -3 $R/gpurun_out/r04_c_tests.log
