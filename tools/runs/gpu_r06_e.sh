#!/bin/bash
# round 6, call E: the tiled VAE with frame-batch stream alternation inside each tile class (A/B), bit-identity of the tiled tests
mkdir -p gpurun_out
O=gpurun_out/r06_e
timeout 900 python tools/tiled_bench.py --reps 3 > ${O}_tiled_bench.log 2>&1
echo "tiled bench exit $?" > ${O}_status.log
timeout 1200 python -m pytest tests/test_e2e_gpu.py -x -q -s -m gpu -k "tiling or tile or long_clip" > ${O}_tiled_tests.log 2>&1
echo "tiled tests exit $?" >> ${O}_status.log
cat ${O}_status.log; tail -3 ${O}_tiled_tests.log
python - <<'PY'
import json
for l in open("gpurun_out/r06_e_tiled_bench.log"):
    if l.startswith("{"):
        r=json.loads(l)
        for k,v in r.items(): print(k, v)
PY
