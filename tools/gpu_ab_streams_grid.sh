#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab.log
for cfg in "1 256" "2 0" "1 0" "2 256"; do set -- $cfg; echo "== streams=$1 grid=$2" >> gpurun_out/ab.log; DOVE_VAE_STREAMS=$1 DOVE_HALO4X_GRID=$2 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.readline()); rf=r['roofline']
print('fps %.2f ms %.1f halo4x %.0f TF share %.2f all_igemm %.0f TF'%(r['value'],r['ms_per_step'],rf['achieved'],rf['share_of_step_time'],rf['all_igemm_kernels']['achieved']))" >> gpurun_out/ab.log; done
cat gpurun_out/ab.log
