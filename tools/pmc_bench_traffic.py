"""Summarise FETCH_SIZE / WRITE_SIZE passes of `rocprofv3 --pmc ... -- python bench.py --steps 1` per kernel and write
profiles/pmc_traffic.json (HBM bytes per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM)."""
import csv
import glob
import json
import re
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dove_amd.lib import kernel_source_sha256  # noqa: E402

d, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
partial = defaultdict(int)          # kernel family -> launches of its PARTIAL-tile instantiations (conv3x3_halo4x_kernel<..., 1 | 2>), FETCH_SIZE pass
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(fn) as f:
        for row in csv.DictReader(f):
            k = re.sub(r"[<(].*", "", row.get("Kernel_Name", "")).replace("void ", "")[:40]
            c = row.get("Counter_Name")
            if c == "FETCH_SIZE" and re.search(r"conv3x3_halo4x_kernel<.*, [12]>", row.get("Kernel_Name", "")):
                partial[k] += 1
            if c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
                a = acc[k][c]
                a[0] += float(row.get("Counter_Value", 0) or 0)
                a[1] += 1
res = {}
for k, v in acc.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v and not k.startswith("at::") and "rocclr" not in k:
        # one conv CALL of the halo kernel is its main launch plus up to two partial-tile launches (round 6): bytes are per call, and
        # `launches` counts calls - what bench.py's profiler hook counts and compares with
        n = v["FETCH_SIZE"][1] - partial.get(k, 0)
        nw = v["WRITE_SIZE"][1] * n / v["FETCH_SIZE"][1]
        fetch = v["FETCH_SIZE"][0] / n * 1024 * 2
        write = v["WRITE_SIZE"][0] / nw * 1024
        res[k] = {"launches": n, "kernel_launches": v["FETCH_SIZE"][1], "fetch_bytes_per_launch_x2corr": fetch, "write_bytes_per_launch": write,
                  "hbm_bytes_per_launch": fetch + write}
# whole-step MFMA-pipe utilisation: busy cycles summed over the 1024 SIMDs / (active cycles per XCD x 1024), over every kernel
mb = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"][0] for v in acc.values() if "SQ_VALU_MFMA_BUSY_CYCLES" in v)
ga = sum(v["GRBM_GUI_ACTIVE"][0] for v in acc.values() if "GRBM_GUI_ACTIVE" in v)
mfma_busy = {k: v["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (v["GRBM_GUI_ACTIVE"][0] / 8 * 1024)
             for k, v in acc.items() if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE", [0])[0] > 0}
top = {"kernel_source_sha256": kernel_source_sha256(),
       "whole_step_mfma_busy_frac": (mb / (ga / 8 * 1024)) if ga else None,
       "mfma_busy_frac_per_kernel": {k: round(x, 4) for k, x in sorted(mfma_busy.items(), key=lambda kv: -kv[1])[:12]},
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants --vae-streams 1 (ONE stream: the TCC counters of a dispatch count whatever the chip moves while it runs)",
       "fetch_correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncorrected", "per_kernel": res}
for dk in ("conv3x3_halo4x_kernel", "conv3x3_halo8_kernel"):
    if dk in res:
        top["dominant_kernel"] = dk
        top["dominant_hbm_bytes_per_launch"] = res[dk]["hbm_bytes_per_launch"]
        break
json.dump(top, open(out, "w"), indent=1)
print("whole-step MFMA-pipe busy fraction:", top["whole_step_mfma_busy_frac"])
print({k: v for k, v in list(top["mfma_busy_frac_per_kernel"].items())[:6]})
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:10]:
    print(f"{k:40s} launches {v['launches']:5d}  HBM/launch {v['hbm_bytes_per_launch']/1e9:8.3f} GB")
