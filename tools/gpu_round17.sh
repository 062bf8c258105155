#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
