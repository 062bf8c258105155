#!/bin/bash
# round 4, call F: heavy-tailed parity stress, mixed softmax paths, attention cutoff tests, piggy-backed score bound at world 2/4/8, bench line with the new variants
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -s -k "heavy or mixed" > gpurun_out/r04_f_parity.log 2>&1; echo "parity rc $?" >> gpurun_out/r04_f_parity.log
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > gpurun_out/r04_f_attn.log 2>&1; echo "attn rc $?" >> gpurun_out/r04_f_attn.log
timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -s -k "bit_identical" > gpurun_out/r04_f_dist.log 2>&1; echo "dist rc $?" >> gpurun_out/r04_f_dist.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r04_f_bench.log 2>&1; echo "bench rc $?" >> gpurun_out/r04_f_bench.log
grep -h "rc \|passed\|failed\|\[heavy\|\[mixed" gpurun_out/r04_f_parity.log gpurun_out/r04_f_attn.log gpurun_out/r04_f_dist.log | tail -30
tail -c 3000 gpurun_out/r04_f_bench.log
