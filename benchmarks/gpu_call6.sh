#!/bin/bash
set -o pipefail
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"
for cfg in "bf16 fwd" "bf16 bwd" "f32 bwd" "f32 fwd"; do
  tag=$(echo $cfg | tr ' ' '_')
  mkdir -p gpurun_out/r2/k1pmc_$tag
  bash benchmarks/run_pmc.sh gpurun_out/r2/k1pmc_$tag benchmarks/k1_pmc.py $cfg -- PASS "$P1" PASS "$P2" || exit $?
  python3 benchmarks/attn_pmc_summary.py gpurun_out/r2/k1pmc_$tag.json --kernels=dwconv7_ln_fwd,dwconv7_ln_bwd_strip gpurun_out/r2/k1pmc_$tag/pass1 gpurun_out/r2/k1pmc_$tag/pass2 > /dev/null 2>&1
done
find gpurun_out/r2 -name "*.db" -delete
