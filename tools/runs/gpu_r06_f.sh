#!/bin/bash
# round 6, call F: the C graph's two-stream VAE (DOVE_OPT_VAE_STREAMS): every graph test, the dist tests that drive the C level, full-size timing
mkdir -p gpurun_out
O=gpurun_out/r06_f
timeout 2400 python -m pytest tests/test_graph_gpu.py -x -q -s -m gpu --durations=8 > ${O}_graph_tests.log 2>&1
echo "graph tests exit $?" > ${O}_status.log
timeout 900 python tools/tiled_bench.py --reps 3 --mode untiled --c-level > ${O}_vae_c_level.log 2>&1
echo "vae c-level timing exit $?" >> ${O}_status.log
timeout 900 python bench.py --steps 8 --warmup 2 > ${O}_bench.log 2>&1
echo "bench exit $?" >> ${O}_status.log
cat ${O}_status.log; grep -h "^\[graph\]\|passed\|failed" ${O}_graph_tests.log | tail -5
python - <<'PY'
import json
for l in open("gpurun_out/r06_f_vae_c_level.log"):
    if l.startswith("{"):
        r=json.loads(l)
        for k,v in r.items(): print(k, v)
for l in open("gpurun_out/r06_f_bench.log"):
    if l.startswith("{"):
        r=json.loads(l); print(round(r["value"],3), round(r["ms_per_step"],2), r["roofline"]["frac"], r["roofline"]["traffic_source"][:80] if r["roofline"]["traffic_source"] else None)
        for v in r.get("variants",[]): print("  ", v["name"][:70], round(v["value"],2), v.get("throughput_vs_untiled_times_flop_ratio"))
PY
