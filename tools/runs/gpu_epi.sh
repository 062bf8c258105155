#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/r03_halo4x_epilogue.log
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" 2>&1 | tail -3 > $L
for c in "128 128" "256 256"; do timeout -k 10 200 python tools/archive/halo4x_timing.py $c 2>&1 | grep "wave 0" >> $L; done
timeout -k 10 400 python tools/e2e_env_ab.py DOVE_IGEMM_ABLATE 64 0 5 2>&1 | grep -v amdgpu.ids | tail -1 >> $L
cat $L
