#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "groupnorm or layernorm or spatial" 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
timeout 600 python tools/microbench.py --only "gn_,ln_mod" > gpurun_out/microbench.log 2>&1
bash tools/gpu_pmc.sh
cat gpurun_out/pytest_gpu.log | tail -3; cat gpurun_out/microbench.log | tail -8; cat gpurun_out/pmc.log
