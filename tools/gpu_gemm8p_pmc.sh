#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/pmc8
i=0
for PMC in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS" \
           "FETCH_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc8/p$i -- python $R/tools/gemm8p_ab.py ff1 > /tmp/pmc8_$i.log 2>&1
  echo "pass $i ($PMC) exit $?" >> $R/gpurun_out/r03_gemm8p_pmc.log
  tail -2 /tmp/pmc8_$i.log >> $R/gpurun_out/r03_gemm8p_pmc.log
done
python $R/tools/pmc_summary.py /tmp/pmc8 $R/gpurun_out/r03_gemm8p_pmc.json >> $R/gpurun_out/r03_gemm8p_pmc.log 2>&1
cat $R/gpurun_out/r03_gemm8p_pmc.log
