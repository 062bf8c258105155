#!/bin/bash
# round 6, call I: the real-size encoder / decoder comparisons with the two oracles side by side
mkdir -p gpurun_out
O=gpurun_out/r06_i
timeout 2400 python -m pytest tests/test_prodshape_gpu.py -x -q -s -m gpu -k "oracle" --durations=5 > ${O}_oracle_tests.log 2>&1
echo "oracle tests exit $?" > ${O}_status.log
cat ${O}_status.log; grep -h "^\[\|^\.\[" ${O}_oracle_tests.log | cut -c1-400; tail -8 ${O}_oracle_tests.log
