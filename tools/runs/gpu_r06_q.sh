#!/bin/bash
# round 6, last call: conv operator tests + the final bundle at the LAST tree (kernel-source hash changed with the kFill guard)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" 2>&1 | tail -2 > gpurun_out/r06_q_ops.log
cat gpurun_out/r06_q_ops.log
bash tools/runs/gpu_r06_final2.sh > gpurun_out/r06_q_bundle.log 2>&1
tail -12 gpurun_out/r06_q_bundle.log
