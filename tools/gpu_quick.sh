#!/bin/bash
# quick confidence bundle after a kernel change: the ops tests named by $1 (pytest -k), the 2-layer e2e gate, a 3-step bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -p no:cacheprovider -k "${1:-attention}" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -x -q -p no:cacheprovider -k "psnr or stage" 2>&1 | tail -3
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1
tail -1 gpurun_out/bench_quick.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('frames/s', round(d['value'],2), 'ms', round(d['ms_per_step'],1), 'halo4x PF', round(d['roofline']['achieved'],1)); [print(' ',k, round(v['ms'],1),'ms', round(v['tflops']),'TF') for k,v in d['roofline']['top_classes'].items()]"
