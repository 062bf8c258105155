#!/bin/bash
# round 6, second session, last call: PMC passes + the driver s bench command + the one-stream rocprofv3 summary at the FINAL tree (ABI 15: tiled-path operators, partial tile column)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash tools/runs/gpu_pmc_bench.sh > /dev/null 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/r06_bench_final2.log
cd /tmp && rm -rf /tmp/prof2 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o r06 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --vae-streams 1 > $R/gpurun_out/r06_prof2_1stream.log 2>&1
cd $R; cp /tmp/prof2/r06_kernel_stats.csv gpurun_out/r06_bench_final2_kernel_stats.csv
python - <<'PY'
import json
for l in open("gpurun_out/r06_bench_final2.log"):
    if l.startswith("{"):
        r=json.loads(l); rf=r["roofline"]
        print(round(r["value"],3), round(r["ms_per_step"],2), round(r["whole_path_tflops_per_gpu"],1), rf["frac"], rf["traffic"], rf["avg_launch_ms"], (rf["traffic_source"] or "")[:90])
        for v in r.get("variants",[]): print("  ", v["name"][:70], round(v["value"],2), v.get("throughput_vs_untiled_times_flop_ratio"), v.get("no_shift_heads_frac"))
        print(r.get("stage_parity",{}).get("passed"), r.get("parity_gate"), r.get("cpu_baseline",{}).get("value"))
PY
head -4 gpurun_out/r06_bench_final2_kernel_stats.csv | cut -c1-150
