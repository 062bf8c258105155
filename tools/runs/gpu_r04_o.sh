#!/bin/bash
# round 4: the whole clip on real and on all-zero operands (schedule-limited time vs power give-back)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 500 python tools/archive/clip_zero_operands.py 6 2>&1 | grep -v amdgpu > gpurun_out/r04_clip_zero_operands.log
cat gpurun_out/r04_clip_zero_operands.log
