#!/bin/bash
# round 6, call H: torch's allocator with expandable segments under the two-stream façade (reserved memory, speed)
mkdir -p gpurun_out
O=gpurun_out/r06_h
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True timeout 600 python bench.py --steps 8 --warmup 2 --no-variants --no-cpu-baseline > ${O}_bench_expandable.log 2>&1
echo "bench expandable exit $?" > ${O}_status.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-variants --no-cpu-baseline > ${O}_bench_default.log 2>&1
echo "bench default exit $?" >> ${O}_status.log
cat ${O}_status.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_h_bench_*.log")):
    ok=False
    for l in open(f):
        if l.startswith("{"):
            r=json.loads(l); ok=True; print(f, round(r["value"],3), round(r["ms_per_step"],2), r.get("hbm_peak_allocated_gb"), r.get("hbm_peak_reserved_gb"))
    if not ok: print(f, open(f).read()[-800:])
PY
