#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -s -p no:cacheprovider -k "mx" 2>&1 | tail -12
timeout 300 python tools/archive/mx_bench.py 2>&1 | tail -12
