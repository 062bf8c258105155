#!/bin/bash
# Round-5 measurement bundle from ONE box at the HEAD of the round: the PMC passes over the bench FIRST (pmc_traffic.json with the kernel-source
# hash, copied into profiles/ on the box so that the bench line of this same bundle replays it), the driver-like bench line, rocprofv3 kernel stats
# of the same command, then the full GPU suite and the smoke test.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash tools/runs/gpu_pmc_bench.sh > /dev/null 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r05_bench.log
cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r05 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $R/gpurun_out/r05_prof.log 2>&1
cd $R; cp /tmp/prof/r05_kernel_stats.csv gpurun_out/r05_bench_kernel_stats.csv
timeout 2400 python -m pytest tests -m gpu -x -q -s -p no:cacheprovider 2>&1 | grep -a "weight sums\]\|\[heavy x4\]\|\[bench single\|\[tol\]\|\[dit42\|passed\|failed\|Error\|error\|assert" | cut -c1-600 > gpurun_out/r05_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > gpurun_out/r05_smoke.log
tail -4 gpurun_out/r05_pytest_gpu.log; cat gpurun_out/r05_smoke.log; tail -1 gpurun_out/r05_bench.log | cut -c1-400; head -6 gpurun_out/r05_bench_kernel_stats.csv | cut -c1-140; tail -12 gpurun_out/pmc_traffic.txt
