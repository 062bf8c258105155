#!/bin/bash
# round 4, call G: C-level configs[2] at real size (8 threads) + the bench's single-clip block through the oversubscribe debug path (2 ranks on GPU 0)
mkdir -p gpurun_out
timeout 900 python tools/clevel_sharded_real.py 8 > gpurun_out/r04_clevel_real.log 2>&1; echo "rc $?" >> gpurun_out/r04_clevel_real.log
timeout 900 python bench.py --gpus 2 --oversubscribe --steps 2 --warmup 1 --layers 2 --no-cpu-baseline > gpurun_out/r04_bench_oversub.log 2>&1; echo "rc $?" >> gpurun_out/r04_bench_oversub.log
tail -3 gpurun_out/r04_clevel_real.log; tail -2 gpurun_out/r04_bench_oversub.log | cut -c1-3000
