#!/bin/bash
# Round-4 measurement bundle from ONE box: the driver's bench line (with the vae_tiling / running-maximum / MXFP8 variants, stage parity), rocprofv3 kernel
# stats of the same command (untiled) and of the tiled variant, the HBM-traffic / MFMA-busy PMC passes over the bench, the tiled VAE A/B.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python bench.py --steps 10 --warmup 3 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r04_bench.log
cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r04 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $R/gpurun_out/r04_prof.log 2>&1
cd $R; cp /tmp/prof/r04_kernel_stats.csv gpurun_out/r04_bench_kernel_stats.csv
timeout 600 python tools/tiled_bench.py --reps 3 --c-level 2>&1 | tail -1 > gpurun_out/r04_tiled_final.log
bash tools/runs/gpu_pmc_bench.sh > /dev/null 2>&1
tail -1 gpurun_out/r04_bench.log | cut -c1-1200; head -12 gpurun_out/r04_bench_kernel_stats.csv | cut -c1-130; cat gpurun_out/r04_tiled_final.log | cut -c1-900; tail -14 gpurun_out/pmc_traffic.txt
