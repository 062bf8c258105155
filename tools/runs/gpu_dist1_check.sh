#!/bin/bash
# 1-rank RCCL run of the sharded single-clip path (the builder's boxes have one GPU): every collective / p2p call site of
# dove_amd.dist goes through backend "nccl"; bit-identity with process_video is asserted by tools/dist_sharded_check.py.  Then the
# same clip at full size through bench.py: plain path vs the sharded path at world 1 (what the sharding machinery costs before any
# communication exists), back to back on one box.
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1"
timeout 600 $L --master-port 29512 tools/dist_sharded_check.py 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/dist1.log
for MODE in "" "--single-clip"; do
  timeout 900 $L --master-port 29513 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-variants $MODE 2>&1 | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench.py (1 rank over RCCL) $MODE:', d['scaling'], round(d['value'],2), 'frames/s', round(d['ms_per_step'],1), 'ms per clip')" | tee -a gpurun_out/dist1.log
done
