#!/bin/bash
# round 4, call D: C-level tile batching (graph.hip) vs the facade: tests + timing at 33x720x1280
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_e2e_gpu.py -x -q -k "tiling or tile_batching or stages_bit_exact or sr_clip" > gpurun_out/r04_d_tests.log 2>&1; echo "tests rc $?" >> gpurun_out/r04_d_tests.log
timeout 600 python tools/tiled_bench.py --reps 3 --c-level > gpurun_out/r04_tiled_clevel.log 2>&1
tail -5 gpurun_out/r04_d_tests.log; tail -2 gpurun_out/r04_tiled_clevel.log
