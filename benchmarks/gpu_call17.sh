#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2/gemm_pmc
bash benchmarks/run_pmc.sh gpurun_out/r2/gemm_pmc benchmarks/gemm_pmc.py -- \
  PASS "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  PASS "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum" \
  PASS "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_VALU_MFMA_COEXEC_CYCLES TCC_EA0_RDREQ_sum TCC_REQ_sum" \
  || exit $?
python3 benchmarks/attn_pmc_summary.py gpurun_out/r2/gemm_pmc.json --kernels=gemm_big_k gpurun_out/r2/gemm_pmc/pass1 gpurun_out/r2/gemm_pmc/pass2 gpurun_out/r2/gemm_pmc/pass3 > /dev/null 2>&1
find gpurun_out/r2 -name "*.db" -delete
