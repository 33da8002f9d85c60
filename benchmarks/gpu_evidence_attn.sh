#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r2/ev
mkdir -p $O/attn
bash benchmarks/run_pmc.sh $O/attn benchmarks/attn_pmc.py -- \
  PASS "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  PASS "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum" \
  PASS "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES TCC_EA0_RDREQ_sum TCC_REQ_sum" \
  || exit $?
python3 benchmarks/attn_pmc_summary.py $O/r02_attn3_pmc.json --kernels=attn_fwd3,attn_bwd_dq3,attn_bwd_dkv3,attn_delta $O/attn/pass1 $O/attn/pass2 $O/attn/pass3 > /dev/null 2>&1
find $O -name "*.db" -delete; rm -rf $O/attn
wc -c $O/r02_attn3_pmc.json
