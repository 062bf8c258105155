#!/bin/bash
# First GPU call of the round after round 4: everything that was prepared on the CPU after the GPU budget ran out.
#  1. K-walk model: what the global -> LDS path costs on top of the product's 16x16x32 step (modes 10 vs 9), on N(0,1) and on zero operands
#  2. phase accounting of the product conv (K walk / pre-epilogue barrier / epilogue / store drain), both walks
#  3. the conv and GEMM A/Bs again on this box (bit-identity + timings)
#  4. flash attention on 16x16x32 (tools/exp/attn16_exp.hip, written without a GPU): correctness against torch + the product kernel, timing
#  5. the bench as the driver runs it (first line with BOTH the conv and the GEMMs on 16x16x32)
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
{
echo "### convalt shape"; timeout 100 python tools/archive/convalt.py shape 4000
echo "### convalt shape zeros"; timeout 100 python tools/archive/convalt.py shape 4000 zeros
echo "### halo4x_timing 16x16x32"; timeout 100 python tools/archive/halo4x_timing.py
echo "### halo4x_timing 32x32x16"; DOVE_HALO_M16=0 timeout 100 python tools/archive/halo4x_timing.py
echo "### halo_m16_ab check time"; timeout 200 python tools/archive/halo_m16_ab.py check time
echo "### gemm_m16_ab"; timeout 100 python tools/archive/gemm_m16_ab.py
echo "### attn16_ab (first run of this kernel on a GPU)"; timeout 150 python tools/archive/attn16_ab.py
} 2>&1 | grep -v amdgpu > gpurun_out/next_first.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/next_bench.log
tail -60 gpurun_out/next_first.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/next_bench.log").read())
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], [round(v["value"],2) for v in d["variants"]], d["stage_parity"]["passed"])
PY
