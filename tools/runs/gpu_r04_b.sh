#!/bin/bash
# round 4, call B: tile batching - bit identity with the loop, tiled tests vs the oracle, timing at 33x720x1280
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_e2e_gpu.py -x -q -k "tiling or tile_batching" > gpurun_out/r04_b_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r04_b_tests.log
timeout 600 python tools/tiled_bench.py --reps 3 > gpurun_out/r04_tiled_batched.log 2>&1
tail -5 gpurun_out/r04_b_tests.log; tail -2 gpurun_out/r04_tiled_batched.log
