#!/bin/bash
# Round-6 measurement bundle from ONE box at the HEAD of the round: the PMC passes over the bench FIRST (one VAE stream: a dispatch's TCC counters
# count whatever the chip moves while it runs; pmc_traffic.json with the kernel-source hash, copied into profiles/ on the box so that the bench line
# of this same bundle replays it), the driver's own bench command, rocprofv3 kernel stats of the bench on one stream (the durations the roofline is
# checked against) and on the product's two streams, then the full GPU suite exactly as the driver runs it (one process) and the smoke test.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash tools/runs/gpu_pmc_bench.sh > /dev/null 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r06_bench.log
cd /tmp && rm -rf /tmp/prof1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o r06 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --vae-streams 1 > $R/gpurun_out/r06_prof_1stream.log 2>&1
cd $R; cp /tmp/prof1/r06_kernel_stats.csv gpurun_out/r06_bench_kernel_stats_1stream.csv
cd /tmp && rm -rf /tmp/prof2 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o r06 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $R/gpurun_out/r06_prof_2streams.log 2>&1
cd $R; cp /tmp/prof2/r06_kernel_stats.csv gpurun_out/r06_bench_kernel_stats_2streams.csv
timeout 3000 python -m pytest tests/ -x -q -m gpu -s -p no:cacheprovider --durations=10 > gpurun_out/r06_pytest_gpu_full.log 2>&1
echo "pytest -m gpu exit $?" > gpurun_out/r06_bundle_status.log
grep -a "weight sums\]\|\[heavy x4\]\|\[bench single\|\[tol\]\|\[dit42\|\[encoder 9x\|\[dit 2 layers\|\[mixed softmax\] score\|\[graph\]\|passed\|failed\|Error\|error" gpurun_out/r06_pytest_gpu_full.log | cut -c1-700 > gpurun_out/r06_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > gpurun_out/r06_smoke.log
cat gpurun_out/r06_bundle_status.log; tail -4 gpurun_out/r06_pytest_gpu.log; cat gpurun_out/r06_smoke.log; tail -1 gpurun_out/r06_bench.log | cut -c1-400; head -6 gpurun_out/r06_bench_kernel_stats_1stream.csv | cut -c1-140; tail -12 gpurun_out/pmc_traffic.txt
