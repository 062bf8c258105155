#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -s -p no:cacheprovider -k "mx" 2>&1 | grep -a "rel RMS\|passed\|failed\|Error\|assert\|off;" | cut -c1-400 | tail -20
timeout 300 python tools/archive/attn_mx_bench.py 2>&1 | tail -8 | tee gpurun_out/attn_mx_bench.log
if [ "$1" = "full" ]; then
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -s -p no:cacheprovider -k "mxfp8" 2>&1 | grep -a "^\[\|\.\[\|passed\|failed\|Error\|assert" | cut -c1-900 | tail -12
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dit-linear mxfp8 --dit-attention mxfp8 > gpurun_out/bench_mxfp8_attn.log 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dit-linear mxfp8 > gpurun_out/bench_mxfp8.log 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1
for f in bench_mxfp8_attn bench_mxfp8 bench_bf16; do tail -1 gpurun_out/$f.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', 'frames/s', round(d['value'],2), 'ms', round(d['ms_per_step'],1), d['dtype'][:60])"; done
fi
