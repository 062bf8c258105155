#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/pmc8
: > $R/gpurun_out/r03_gemm8p_pmc.log
for CASE in qkv "ff1 shape, plain"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d "/tmp/pmc8/${CASE// /_}" -- python $R/tools/archive/gemm8p_ab.py "$CASE" > /tmp/pmc8.log 2>&1
  echo "== $CASE (rocprofv3 exit $?)" >> $R/gpurun_out/r03_gemm8p_pmc.log
  python $R/tools/archive/gemm8p_pmc.py "/tmp/pmc8/${CASE// /_}" >> $R/gpurun_out/r03_gemm8p_pmc.log 2>&1
done
cat $R/gpurun_out/r03_gemm8p_pmc.log
