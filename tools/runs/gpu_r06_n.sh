#!/bin/bash
# round 6, call N: frame-batches of a tile class on 2 / 3 / 4 streams (the deep levels' launches are 1.5 - 2.25 rounds of one tile)
mkdir -p gpurun_out
timeout 900 python tools/tiled_bench.py --reps 3 > gpurun_out/r06_n_tiled_bench.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r06_n_tiled_bench.log"):
    if l.startswith("{"):
        r=json.loads(l)
        for k,v in r.items(): print(k, v)
PY
timeout 600 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu -k "tiling or tile" 2>&1 | tail -2
