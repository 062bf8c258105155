#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab.log
for X in 1 0 1 0; do echo "== DOVE_CONV_GN_FUSE=$X" >> gpurun_out/ab.log; DOVE_CONV_GN_FUSE=$X timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.readline()); rf=r['roofline']
print('fps %.2f ms %.1f dom %.0f TF share %.2f all_igemm %.0f TF'%(r['value'],r['ms_per_step'],rf['achieved'],rf['share_of_step_time'],rf['all_igemm_kernels']['achieved']))
for k,v in list(rf['top_classes'].items())[:2]: print('   %-55s %8.2f ms %7.1f TF %4d launches'%(k,v['ms'],v['tflops'],v['launches']))" >> gpurun_out/ab.log; done
cat gpurun_out/ab.log
