#!/bin/bash
# round 3, first GPU call: the new tests (sharded mode on one GPU, production-shape samples, north-star tolerance, tiling at the
# real geometry), the operator / graph suites touched by the ABI 10 changes, the bench's multi-rank path (debug, 2 ranks on GPU 0)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
{
  echo "== test_dist_gpu"; timeout 900 python -m pytest tests/test_dist_gpu.py -q -s 2>&1 | grep -v amdgpu | tail -25
  echo "== prodshape"; timeout 600 python -m pytest tests/test_ops_gpu.py -q -s -k prodshape 2>&1 | tail -15
  echo "== tolerance"; timeout 900 python -m pytest tests/test_parity_gpu.py -q -s -k "north_star or stagewise" 2>&1 | tail -15
  echo "== tiling"; timeout 600 python -m pytest tests/test_e2e_gpu.py -q -s -k tiling 2>&1 | tail -12
  echo "== ops + graph + abi"; timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_graph_gpu.py tests/test_abi.py -q -x -k "not prodshape" 2>&1 | tail -8
  echo "== bench oversubscribe x2"; timeout 600 python bench.py --gpus 2 --oversubscribe --steps 1 --warmup 1 --layers 2 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -3 | cut -c1-3000
} > gpurun_out/r03_a.log 2>&1
tail -c 6000 gpurun_out/r03_a.log
