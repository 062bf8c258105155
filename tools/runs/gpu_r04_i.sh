#!/bin/bash
# round 4, call I: sub-pixel form of the upsample-fused convs (w_sub): conv tests, e2e / bit-identity suites, within-run A/B
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_graph_gpu.py tests/test_e2e_gpu.py -x -q -k "conv or stage or stages or sr_clip or tiling or tile_batching or two_stream or one_clip" > gpurun_out/r04_i_tests.log 2>&1; echo "rc $?" >> gpurun_out/r04_i_tests.log
timeout 600 python tools/archive/first_frame_ab.py sub > gpurun_out/r04_subpixel_ab.log 2>&1
tail -25 gpurun_out/r04_i_tests.log; cat gpurun_out/r04_subpixel_ab.log | tail -5
