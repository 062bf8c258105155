"""Per-kernel MFMA-pipe busy fraction and shader clock of gemm4x / gemm8p from one rocprofv3 --kernel-trace --pmc run directory
(SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE [, SQ_WAIT_INST_ANY, SQ_WAVE_CYCLES, SQ_BUSY_CYCLES]).  Usage: gemm8p_pmc.py <dir>"""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
dur = defaultdict(list)
for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        dur[(re.sub(r"\(.*", "", r["Kernel_Name"])[:48], r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
cnt = defaultdict(lambda: defaultdict(list))
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:48]
        if "gemm" not in k:
            continue
        cnt[(k, r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
agg = defaultdict(lambda: defaultdict(list))
for (k, did), c in cnt.items():
    if (k, did) not in dur:
        continue
    ns = dur[(k, did)]
    key = (k, round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1e8))      # separates the shapes by their MFMA work
    agg[key]["ns"].append(ns)
    for n, v in c.items():
        agg[key][n].append(v)
for key in sorted(agg):
    a = {n: sum(v) / len(v) for n, v in agg[key].items()}
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    print(f"{key[0]:46s} work~{key[1]:3d}e8  {a['ns'] / 1e3:8.1f} us  clock {cyc / a['ns']:.3f} GHz  MFMA busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}"
          + "".join(f"  {n} {a[n] / (cyc * 1024):.3f}" for n in ("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM") if n in a))
