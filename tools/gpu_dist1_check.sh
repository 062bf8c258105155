#!/bin/bash
# 1-rank RCCL run of the sharded single-clip path (the builder's boxes have one GPU): every collective / p2p call site of
# dove_amd.dist goes through backend "nccl"; bit-identity with process_video is asserted by tools/dist_sharded_check.py.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 tools/dist_sharded_check.py 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/dist1.log
timeout 900 python bench.py --gpus 1 --steps 2 --warmup 1 --single-clip --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700 | tee -a gpurun_out/dist1.log
