#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/all.log 2>&1; rc=$?; tail -5 gpurun_out/r2/all.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 60 rocprofv3 -L > gpurun_out/r2/counters.txt 2>&1
mkdir -p gpurun_out/r2/attn_pmc
bash benchmarks/run_pmc.sh gpurun_out/r2/attn_pmc benchmarks/attn_pmc.py -- \
  PASS "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  PASS "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  PASS "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
  || exit $?
export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/r2/attn_trace -o t -- python3 benchmarks/attn_pmc.py > gpurun_out/r2/attn_trace.log 2>&1
python3 benchmarks/attn_pmc_summary.py gpurun_out/r2/attn_pmc.json gpurun_out/r2/attn_pmc/pass1 gpurun_out/r2/attn_pmc/pass2 gpurun_out/r2/attn_pmc/pass3 > gpurun_out/r2/attn_pmc_summary.log 2>&1
for f in gpurun_out/r2/attn_pmc/*.log; do tail -n 3 "$f"; done
# keep what comes back small: the per-dispatch CSVs are large
find gpurun_out/r2 -name "*.db" -delete
