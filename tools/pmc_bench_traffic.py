"""Summarise FETCH_SIZE / WRITE_SIZE passes of `rocprofv3 --pmc ... -- python bench.py --steps 1` per kernel and write
profiles/pmc_traffic.json (HBM bytes per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(fn) as f:
        for row in csv.DictReader(f):
            k = re.sub(r"[<(].*", "", row.get("Kernel_Name", "")).replace("void ", "")[:40]
            c = row.get("Counter_Name")
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                a = acc[k][c]
                a[0] += float(row.get("Counter_Value", 0) or 0)
                a[1] += 1
res = {}
for k, v in acc.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v and not k.startswith("at::") and "rocclr" not in k:
        n = v["FETCH_SIZE"][1]
        fetch = v["FETCH_SIZE"][0] / n * 1024 * 2
        write = v["WRITE_SIZE"][0] / v["WRITE_SIZE"][1] * 1024
        res[k] = {"launches": n, "fetch_bytes_per_launch_x2corr": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write}
top = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline",
       "fetch_correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncorrected", "per_kernel": res}
for dk in ("conv3x3_halo4x_kernel", "conv3x3_halo8_kernel"):
    if dk in res:
        top["dominant_kernel"] = dk
        top["dominant_hbm_bytes_per_launch"] = res[dk]["hbm_bytes_per_launch"]
        break
json.dump(top, open(out, "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:10]:
    print(f"{k:40s} launches {v['launches']:5d}  HBM/launch {v['hbm_bytes_per_launch']/1e9:8.3f} GB")
