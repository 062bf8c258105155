#!/bin/bash
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_t5_gpu.py -m gpu -x -q -s -p no:cacheprovider 2>&1 | grep -a "^\[\|\.\[\|passed\|failed\|Error\|assert\|error" | cut -c1-400 | tail -12
