#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ablate.log
for c in "128 128" "256 256"; do DOVE_HALO4X_CFG=9 DOVE_CONV_HALO4X=1 timeout 300 python tools/halo4x_timing.py $c 2>&1 | grep -v amdgpu.ids >> gpurun_out/ablate.log; done
cat gpurun_out/ablate.log
