#!/bin/bash
# round 6, call G: C graph two-stream VAE with the arena split per stream (A/B 1 vs 2 streams in one process), graph tests, the bench's variants again
mkdir -p gpurun_out
O=gpurun_out/r06_g
timeout 900 python tools/tiled_bench.py --reps 3 --mode both --c-level > ${O}_vae_c_level.log 2>&1
echo "vae c-level timing exit $?" > ${O}_status.log
timeout 2400 python -m pytest tests/test_graph_gpu.py -x -q -s -m gpu --durations=5 > ${O}_graph_tests.log 2>&1
echo "graph tests exit $?" >> ${O}_status.log
timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > ${O}_bench.log 2>&1
echo "bench exit $?" >> ${O}_status.log
cat ${O}_status.log; grep -h "^\[graph\]\|passed\|failed" ${O}_graph_tests.log | tail -5
python - <<'PY'
import json
for l in open("gpurun_out/r06_g_vae_c_level.log"):
    if l.startswith("{"):
        r=json.loads(l)
        for k,v in r.items(): print(k, v)
for l in open("gpurun_out/r06_g_bench.log"):
    if l.startswith("{"):
        r=json.loads(l); print(round(r["value"],3), round(r["ms_per_step"],2), r["roofline"]["frac"], r.get("hbm_peak_reserved_gb"))
        for v in r.get("variants",[]): print("  ", v["name"][:70], round(v["value"],2), v.get("throughput_vs_untiled_times_flop_ratio"))
PY
