#!/bin/bash
# round 6, call B: the full GPU suite at this HEAD (new C-level halo streams, attention dispatch, production-shape tests) + the stream-count A/B
mkdir -p gpurun_out
O=gpurun_out/r06_b
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 > ${O}_pytest_gpu.log 2>&1
echo "pytest gpu exit $?" > ${O}_status.log
for n in 3 4 2; do
  timeout 600 python bench.py --steps 8 --warmup 2 --vae-streams $n --no-variants --no-cpu-baseline > ${O}_bench_${n}streams.log 2>&1
  echo "bench $n streams exit $?" >> ${O}_status.log
done
cat ${O}_status.log
tail -n 25 ${O}_pytest_gpu.log
