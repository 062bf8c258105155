#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_graph_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_q.log 2>&1
tail -1 gpurun_out/bench_q.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('frames/s', round(d['value'],2), 'ms', round(d['ms_per_step'],1))"
cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o q -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; cp /tmp/prof/q_kernel_stats.csv gpurun_out/kernel_stats_q.csv; grep -a "smallk\|qkv_post\|gn_apply\|gemm4x" gpurun_out/kernel_stats_q.csv | cut -c1-60,150-260
