#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "prodshape" 2>&1 | tail -5 | tee gpurun_out/r03_prodshape_mx.log
