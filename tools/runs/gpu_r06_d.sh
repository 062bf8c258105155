#!/bin/bash
# round 6, call D: the new DiT-at-N=18226-vs-oracle test; events in / out of the timed region (A/B on one box)
mkdir -p gpurun_out
O=gpurun_out/r06_d
timeout 1200 python -m pytest tests/test_prodshape_gpu.py -x -q -s -m gpu -k "dit_2_layers" > ${O}_dit_oracle.log 2>&1
echo "dit oracle test exit $?" > ${O}_status.log
for rep in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 2 --no-variants --no-cpu-baseline > ${O}_bench_noevents_$rep.log 2>&1
  echo "bench (no events in the timed region) $rep exit $?" >> ${O}_status.log
  timeout 600 python bench.py --steps 10 --warmup 2 --no-variants --no-cpu-baseline --timed-region-events > ${O}_bench_events_$rep.log 2>&1
  echo "bench (events) $rep exit $?" >> ${O}_status.log
done
cat ${O}_status.log
grep -h "^\[" ${O}_dit_oracle.log | tail -3
tail -3 ${O}_dit_oracle.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_d_bench_*.log")):
    for l in open(f):
        if l.startswith("{"):
            r=json.loads(l); print(f, round(r["value"],3), round(r["ms_per_step"],2), r["roofline"]["frac"])
PY
