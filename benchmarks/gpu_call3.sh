#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2/attn_pmc
bash benchmarks/run_pmc.sh gpurun_out/r2/attn_pmc benchmarks/attn_pmc.py -- \
  PASS "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  PASS "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  PASS "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
  PASS "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU" \
  || exit $?
export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/r2/attn_trace -o t -- python3 benchmarks/attn_pmc.py > gpurun_out/r2/attn_trace.log 2>&1
python3 benchmarks/attn_pmc_summary.py gpurun_out/r2/attn_pmc.json gpurun_out/r2/attn_pmc/pass1 gpurun_out/r2/attn_pmc/pass2 gpurun_out/r2/attn_pmc/pass3 gpurun_out/r2/attn_pmc/pass4 > gpurun_out/r2/attn_pmc_summary.log 2>&1
tail -n 3 gpurun_out/r2/attn_pmc_summary.log
find gpurun_out/r2 -name "*.db" -delete
ls -R gpurun_out/r2 | head -50
