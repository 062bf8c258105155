#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_graph_gpu.py -m gpu -x -q -p no:cacheprovider -k "conv_out or stages or sr_clip_equals or sharded" 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -5
cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o q -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_q.log 2>&1
cd $GRAFT_REPO_ROOT; cp /tmp/prof/q_kernel_stats.csv gpurun_out/kernel_stats_q.csv
grep -a '"metric"' gpurun_out/bench_q.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('frames/s', round(d['value'],2), 'ms', round(d['ms_per_step'],1))"
grep -a "igemm_fast\|conv_out_gather\|ncthw_from_cl" gpurun_out/kernel_stats_q.csv | awk -F'",' '{print substr($1,1,50), $2}'
