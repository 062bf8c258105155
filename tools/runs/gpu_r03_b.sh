#!/bin/bash
# round 3, second GPU call: graph-level options (tiling / MXFP8 / noise_step through the C graph, arena guard), the bench's
# multi-rank path (debug: 2 ranks on GPU 0 over gloo) with the whole JSON line kept, the N = 1 bench line with the MXFP8 variants entry
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
{
  echo "== test_graph_gpu"; timeout 1200 python -m pytest tests/test_graph_gpu.py -q -s -x 2>&1 | grep -v amdgpu | tail -25
  echo "== bench oversubscribe x2"; timeout 600 python bench.py --gpus 2 --oversubscribe --steps 1 --warmup 1 --layers 2 --no-cpu-baseline 2>&1 | grep -v "amdgpu\|Gloo" | tail -2 > gpurun_out/r03_bench_oversub2.log; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r03_bench_oversub2.log') if x.startswith('{')]
d=json.loads(l[-1]); print(json.dumps(d.get('single_clip'), indent=1)); print({k: d[k] for k in ('value','n_gpus','invalid','scaling')})
PY
  echo "== bench N=1"; timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r03_bench_v1.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_v1.log').read())
print({k: d[k] for k in ('value','ms_per_step','psnr_vs_oracle_db')}); print(d['roofline']['achieved'], d['roofline']['frac']); print(json.dumps(d.get('variants'), indent=1)); print(d['cpu_baseline'])
PY
} > gpurun_out/r03_b.log 2>&1
tail -c 5000 gpurun_out/r03_b.log
