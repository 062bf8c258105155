#!/bin/bash
# Round-5 measurement bundle from ONE box at the HEAD of the round: the driver-like bench line, rocprofv3 kernel stats of the same command, the
# HBM-traffic / MFMA-busy PMC passes over the bench (pmc_traffic.json carries the kernel-source hash bench.py checks), the tests added late.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_graph_gpu.py tests/test_ops_gpu.py -x -q -s -m gpu -p no:cacheprovider -k "weight_sums or w_first_sampled or tdup" 2>&1 | grep -a "weight sums\|passed\|failed\|Error\|assert" | cut -c1-500 > gpurun_out/r05_bundle_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r05_bench.log
cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r05 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $R/gpurun_out/r05_prof.log 2>&1
cd $R; cp /tmp/prof/r05_kernel_stats.csv gpurun_out/r05_bench_kernel_stats.csv
bash tools/runs/gpu_pmc_bench.sh > /dev/null 2>&1
cat gpurun_out/r05_bundle_tests.log; tail -1 gpurun_out/r05_bench.log | cut -c1-500; head -12 gpurun_out/r05_bench_kernel_stats.csv | cut -c1-140; tail -14 gpurun_out/pmc_traffic.txt
