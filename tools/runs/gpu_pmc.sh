#!/bin/bash
# PMC passes (counters only, with --kernel-trace) on the dominant kernels at headline shapes.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ONLY=${1:-"conv3d 128->128,attention,linear ff1"}
cd /tmp
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc/p$i -- python $R/tools/microbench.py --only "$ONLY" --iters 2 > /tmp/pmc_$i.log 2>&1
  echo "pass $i exit $?" >> $R/gpurun_out/pmc.log
done
python $R/tools/pmc_summary.py /tmp/pmc $R/gpurun_out/pmc_summary.json > $R/gpurun_out/pmc_summary.txt 2>&1
tail -3 /tmp/pmc_1.log >> $R/gpurun_out/pmc.log
