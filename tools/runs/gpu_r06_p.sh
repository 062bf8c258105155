#!/bin/bash
# round 6, call P: the final bundle again (igemm.hip gained the timing-only kFill parameter after bundle final2: the kernel-source hash changed)
# + the two real-size stage tests with the bf16 yardstick measured LIVE beside the fp32 oracle (DOVE_TEST_BF16_YARDSTICK=1)
bash tools/runs/gpu_r06_final2.sh > gpurun_out/r06_p_bundle.log 2>&1
tail -12 gpurun_out/r06_p_bundle.log
DOVE_TEST_BF16_YARDSTICK=1 timeout 1500 python -m pytest tests/test_prodshape_gpu.py -x -q -s -m gpu -k "vs_oracle_9x720x1280" 2>&1 | grep -v amdgpu > gpurun_out/r06_p_live_yardstick.log
tail -5 gpurun_out/r06_p_live_yardstick.log
