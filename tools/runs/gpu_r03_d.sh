#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_abi.py -x -q -m gpu -k "linear or lin or prodshape or dispatch or gemm" 2>&1 | tail -5 | tee gpurun_out/r03_gemm8p_tests.log
timeout -k 10 150 python tools/archive/gemm8p_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_gemm8p_ab.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants 2>&1 | tail -1 | tee gpurun_out/r03_bench_gemm8p.log
