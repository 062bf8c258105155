#!/bin/bash
# round 4: full GPU suite (as the driver runs it) + smoke
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=15 2>&1 | grep -v "amdgpu\|Gloo\|socket.cpp" | tail -40 > gpurun_out/r04_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/r04_smoke.log
cat gpurun_out/r04_pytest_gpu.log gpurun_out/r04_smoke.log
