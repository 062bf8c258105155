#!/bin/bash
# Round-3 measurement bundle: full GPU test suite, smoke, bench line (with the MXFP8 `variants` entry), rocprofv3 kernel stats of the
# same command, PMC passes on the dominant kernels (microbench shapes) and HBM-traffic / MFMA-busy passes over the bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v "amdgpu\|Gloo\|socket.cpp" | tail -6 > gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/smoke.log
timeout 900 python bench.py --steps 10 --warmup 3 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/bench.log
cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r03 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $R/gpurun_out/prof.log 2>&1
cd $R; cp /tmp/prof/r03_kernel_stats.csv gpurun_out/kernel_stats.csv
bash tools/runs/gpu_pmc.sh "conv3d 128->128,attention,linear ff1"
bash tools/runs/gpu_pmc_bench.sh > /dev/null 2>&1
cat gpurun_out/pytest_gpu.log gpurun_out/smoke.log; tail -1 gpurun_out/bench.log | cut -c1-1500; head -14 gpurun_out/kernel_stats.csv | cut -c1-130; cat gpurun_out/pmc_traffic.txt | tail -20
