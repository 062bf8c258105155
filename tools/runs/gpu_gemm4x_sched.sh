#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in 0 1; do DOVE_GEMM4X_SCHED=$v timeout 200 python tools/archive/gemm4x_sched.py 2>&1 | grep "DOVE_GEMM4X" ; done; done | tee gpurun_out/r03_gemm4x_sched.log
