#!/bin/bash
# round 4: which part of ff2 / out does not speed up on zero operands - the gemm8p main launch or the igemm_fast row tail?
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
{
for data in normal zeros; do
  for only in "linear ff2" "linear out" "linear ff1"; do
    rm -rf /tmp/kt
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/tools/microbench.py --only "$only" --iters 10 --data $data > /tmp/kt.log 2>&1
    echo "# $only, data = $data"
    grep "linear" /tmp/kt.log
    f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
    python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm8p" in r["Name"] or "igemm_fast" in r["Name"]:
        print(f'   {r["Name"][:60]:60s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
  done
done
} > $R/gpurun_out/r04_gemm_tail_split.log 2>&1
cat $R/gpurun_out/r04_gemm_tail_split.log
