#!/bin/bash
# round 6, call C: the rest of the GPU suite (from test_graph_gpu on) after the workspace fix of test_sr_clip_full_size_timing
mkdir -p gpurun_out
O=gpurun_out/r06_c
timeout 3000 python -m pytest tests/test_graph_gpu.py tests/test_ops_gpu.py tests/test_parity_gpu.py tests/test_prodshape_gpu.py tests/test_t5_gpu.py -x -q -s -m gpu --durations=12 > ${O}_pytest_gpu.log 2>&1
echo "pytest gpu (graph..t5) exit $?" > ${O}_status.log
cat ${O}_status.log
grep -n "^\[" ${O}_pytest_gpu.log | tail -40
tail -n 20 ${O}_pytest_gpu.log
