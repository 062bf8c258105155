#!/bin/bash
# round 6, call A: host facts, the new / changed attention + production-shape tests, the two-stream A/B on the bench
mkdir -p gpurun_out
O=gpurun_out/r06_a
{ free -g; nproc; rocm-smi --showmeminfo vram | head -8; } > ${O}_host.log 2>&1
timeout 1500 python -m pytest tests/test_prodshape_gpu.py -x -q -s -m gpu > ${O}_prodshape.log 2>&1
echo "prodshape exit $?" >> ${O}_status.log
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -s -m gpu -k "attention" > ${O}_attn.log 2>&1
echo "ops attention exit $?" >> ${O}_status.log
timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q -s -m gpu -k "mixed_softmax or dit_42" > ${O}_mixed.log 2>&1
echo "mixed exit $?" >> ${O}_status.log
timeout 600 python bench.py --steps 8 --warmup 2 --vae-streams 1 --no-variants --no-cpu-baseline > ${O}_bench_1stream.log 2>&1
echo "bench 1 stream exit $?" >> ${O}_status.log
timeout 900 python bench.py --steps 8 --warmup 2 > ${O}_bench_2stream.log 2>&1
echo "bench 2 streams exit $?" >> ${O}_status.log
cat ${O}_status.log
tail -3 ${O}_prodshape.log ${O}_attn.log ${O}_mixed.log
