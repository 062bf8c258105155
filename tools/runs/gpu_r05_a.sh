#!/bin/bash
# Round-5 bundle A (first GPU call): the new parity tests (multi-seed heavy-tailed gate, production-shape w_first check, bench self-validation),
# then the HEAD evidence bundle - driver-like bench line, rocprofv3 kernel stats of the same command, the three PMC passes with the source hash -
# and the fair attention A/B (16x16x32 experiment kernel vs the product, both on their constant-shift and running-maximum paths).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py tests/test_dist_gpu.py tests/test_graph_gpu.py -x -q -s -m gpu -p no:cacheprovider \
  -k "heavy_tailed or w_first_sampled or bench_single_clip or qkv_post or sharded_on_hip or one_clip_sharded or halo_exchange_c_level" 2>&1 \
  | grep -a "^\[heavy\|^\[weight sums\|^\[bench\|passed\|failed\|Error\|error\|assert" | cut -c1-700 > gpurun_out/r05_a_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r05_bench_head.log
cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r05 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants > $R/gpurun_out/r05_prof.log 2>&1
cd $R; cp /tmp/prof/r05_kernel_stats.csv gpurun_out/r05_bench_kernel_stats.csv
bash tools/runs/gpu_pmc_bench.sh > /dev/null 2>&1
timeout 300 python tools/archive/attn16_ab.py 2>&1 | grep -v amdgpu > gpurun_out/r05_attn16_fair.log
cat gpurun_out/r05_a_tests.log; tail -1 gpurun_out/r05_bench_head.log | cut -c1-600; head -8 gpurun_out/r05_bench_kernel_stats.csv | cut -c1-130; tail -14 gpurun_out/pmc_traffic.txt; cat gpurun_out/r05_attn16_fair.log
