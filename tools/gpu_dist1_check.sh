#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "test_conv" 2>&1 | tail -2
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 2>&1 | tail -2 | cut -c1-400
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 tools/dist_sharded_check.py 2>&1 | grep -v amdgpu | tail -3
