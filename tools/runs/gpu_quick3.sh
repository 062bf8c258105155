#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_graph_gpu.py tests/test_e2e_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4
timeout 300 python tools/archive/attn_mx_bench.py 2>&1 | grep -a "attention\|qkv_post" | tee gpurun_out/attn_q3.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_q.log 2>&1
tail -1 gpurun_out/bench_q.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('frames/s', round(d['value'],2), 'ms', round(d['ms_per_step'],1))"
