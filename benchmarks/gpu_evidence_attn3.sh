#!/bin/bash
# round 3: counters of the attention kernels at the config shape (d = 100) and at head_dim 512, with residency figures
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3/ev
mkdir -p $O
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum"
P3="SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES TCC_EA0_RDREQ_sum TCC_REQ_sum"
IFS=";" read -ra SH <<< "${SHAPES:-32 16 100;32 4 512}"; unset IFS
for shape in "${SH[@]}"; do
  set -- $shape
  tag=d$3
  mkdir -p $O/$tag
  bash benchmarks/run_pmc.sh $O/$tag benchmarks/attn_pmc.py $1 $2 $3 -- PASS "$P1" PASS "$P2" PASS "$P3" || exit $?
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag/trace -o t -- python3 benchmarks/attn_pmc.py $1 $2 $3 > $O/$tag/trace.log 2>&1 || exit $?
  if [ $3 = 512 ]; then
    python3 benchmarks/attn_pmc_summary.py $O/r03_attn_${tag}_pmc.json --kernels=attn_fwd_w,attn_bwd_dq_w,attn_bwd_dkv_w,attn_delta \
      --lds=attn_fwd_w:135168,attn_bwd_dq_w:135168,attn_bwd_dkv_w:137728 $O/$tag/pass1 $O/$tag/pass2 $O/$tag/pass3 > /dev/null 2>&1
  else
    python3 benchmarks/attn_pmc_summary.py $O/r03_attn_${tag}_pmc.json --kernels=attn_fwd_r,attn_bwd_dq_r,attn_bwd_dkv_r,attn_delta \
      --lds=attn_fwd_r:143584,attn_bwd_dq_r:143584,attn_bwd_dkv_r:146144 $O/$tag/pass1 $O/$tag/pass2 $O/$tag/pass3 > /dev/null 2>&1
  fi
  f=$(find $O/$tag/trace -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -E "attn_|Name" "$f" | cut -c1-200 > $O/r03_attn_${tag}_kernel_stats.csv
  find $O/$tag -name "*.db" -delete; rm -rf $O/$tag/pass* $O/$tag/trace
done
ls -la $O
