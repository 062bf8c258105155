#!/bin/bash
# round 3: the whole GPU suite + the default bench line (what the driver runs at round end)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "amdgpu\|Gloo\|socket.cpp" | tail -25 > gpurun_out/r03_pytest_gpu.log
tail -6 gpurun_out/r03_pytest_gpu.log
timeout 900 python bench.py --steps 10 --warmup 3 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r03_bench_v2.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_v2.log').read())
print({k: d[k] for k in ('value','ms_per_step','psnr_vs_oracle_db')}, d['roofline']['achieved'], [ (v['value'], v['speedup_vs_headline_this_run']) for v in d.get('variants',[])])
for k,v in d['roofline']['top_classes'].items(): print(k, v)
PY
