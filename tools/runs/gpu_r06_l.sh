#!/bin/bash
# round 6, call L: partial-tile launches of conv3x3_halo4x (kPart): operator tests (bit-identity against the full-tile form), the VAE-level
# tests, then tiled / untiled VAE timing and the one-stream per-class table
mkdir -p gpurun_out
O=gpurun_out/r06_l
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" > ${O}_ops.log 2>&1
echo "ops(conv) exit $?" > ${O}_status.log
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_graph_gpu.py -x -q -m gpu -k "tiling or tile or two_stream or bit" > ${O}_vae_tests.log 2>&1
echo "vae tests exit $?" >> ${O}_status.log
timeout 900 python tools/tiled_bench.py --reps 3 > ${O}_tiled_bench.log 2>&1
echo "tiled bench exit $?" >> ${O}_status.log
for m in untiled tiled; do timeout 600 python tools/vae_mode_profile.py --mode $m > ${O}_classes_$m.log 2>&1; done
cat ${O}_status.log; tail -3 ${O}_ops.log; tail -3 ${O}_vae_tests.log
python - <<'PY'
import json
for l in open("gpurun_out/r06_l_tiled_bench.log"):
    if l.startswith("{"):
        r=json.loads(l)
        for k,v in r.items(): print(k, v)
PY
head -8 gpurun_out/r06_l_classes_untiled.log | grep -v amdgpu; head -8 gpurun_out/r06_l_classes_tiled.log | grep -v amdgpu
