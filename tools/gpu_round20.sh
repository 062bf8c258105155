#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun.log 2>&1; echo "torchrun exit $?" >> gpurun_out/bench_torchrun.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
tail -3 gpurun_out/bench_torchrun.log | cut -c1-400; tail -2 gpurun_out/smoke.log
