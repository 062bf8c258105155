#!/bin/bash
# round 6, call K: conv_in / conv_out in their split forms inside spatial tiles (dove_tile_gather_bf16, dove_conv_out_gather_cl; ABI 15):
# the new operator tests, every tiled test (facade vs oracle, loop vs batched, C level vs facade), then the tiled VAE timing
mkdir -p gpurun_out
O=gpurun_out/r06_k
nproc > ${O}_host.log; free -g >> ${O}_host.log
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "tile_gather or conv_out_gather or blend or conv_out" > ${O}_ops.log 2>&1
echo "ops exit $?" > ${O}_status.log
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_graph_gpu.py -x -q -s -m gpu -k "tiling or tile or long_clip" > ${O}_tiled_tests.log 2>&1
echo "tiled tests exit $?" >> ${O}_status.log
timeout 900 python tools/tiled_bench.py --reps 3 --c-level > ${O}_tiled_bench.log 2>&1
echo "tiled bench exit $?" >> ${O}_status.log
cat ${O}_status.log; tail -3 ${O}_ops.log; tail -3 ${O}_tiled_tests.log
python - <<'PY'
import json
for l in open("gpurun_out/r06_k_tiled_bench.log"):
    if l.startswith("{"):
        r=json.loads(l)
        for k,v in r.items(): print(k, v)
PY
