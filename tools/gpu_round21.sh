#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
DOVE_VAE_STREAMS=2 timeout 900 python -m pytest tests/test_e2e_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/pytest_gpu.log
rm -f gpurun_out/streams.log
for S in 1 2 1 2; do echo "== STREAMS=$S" >> gpurun_out/streams.log; DOVE_VAE_STREAMS=$S timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "frames/s".\{0,80\}' >> gpurun_out/streams.log; done
tail -4 gpurun_out/pytest_gpu.log; cat gpurun_out/streams.log
