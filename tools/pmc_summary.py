"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv files under a directory) per kernel:
mean counter value per dispatch.  Usage: python tools/pmc_summary.py <dir> <out.json>"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def main(d, out):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                k = re.sub(r"\(.*", "", row.get("Kernel_Name", ""))[:60]
                if "at::native" in k or "rocclr" in k:
                    continue
                c = row.get("Counter_Name")
                v = float(row.get("Counter_Value", 0) or 0)
                a = acc[k][c]
                a[0] += v
                a[1] += 1
                for extra in ("VGPR_Count", "Accum_VGPR_Count", "LDS_Block_Size", "SGPR_Count", "Grid_Size", "Workgroup_Size"):
                    if extra in row and row[extra] not in (None, ""):
                        acc[k]["_" + extra][0] = float(row[extra])
                        acc[k]["_" + extra][1] = 1
    res = {k: {c: a[0] / max(a[1], 1) for c, a in v.items()} | {"_dispatches": max(a[1] for a in v.values())} for k, v in acc.items()}
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for k, v in res.items():
        print(k)
        for c in sorted(v):
            print(f"   {c:36s} {v[c]:.6g}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
