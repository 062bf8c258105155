#!/bin/bash
# round 6: the whole GPU suite in ONE process, exactly as the driver runs it, + smoke, at the final tree
mkdir -p gpurun_out
timeout 3300 python -m pytest tests/ -x -q -m gpu -s -p no:cacheprovider --durations=12 > gpurun_out/r06_suite_full.log 2>&1
echo "pytest -m gpu exit $?" > gpurun_out/r06_suite_status.log
grep -a "\[encoder 9x\|\[decoder 3x\|\[dit 2 layers\|\[graph\]\|passed\|failed\|Error\|error" gpurun_out/r06_suite_full.log | cut -c1-500 > gpurun_out/r06_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r06_suite.log
cat gpurun_out/r06_suite_status.log; cat gpurun_out/r06_suite.log; tail -16 gpurun_out/r06_suite_full.log
