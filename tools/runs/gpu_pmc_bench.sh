#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/pmcb
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "/tmp/pmcb/${PMC// /_}" -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-variants --vae-streams 1 > "/tmp/pmcb_${PMC// /_}.log" 2>&1
  echo "$PMC exit $?" >> $R/gpurun_out/pmc_bench.log
done
python $R/tools/pmc_bench_traffic.py /tmp/pmcb $R/gpurun_out/pmc_traffic.json > $R/gpurun_out/pmc_traffic.txt 2>&1
cat $R/gpurun_out/pmc_traffic.txt
