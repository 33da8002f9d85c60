#!/bin/bash
# Round-2 evidence: everything profiles/ holds for this round comes from this one script (one gpurun call).
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r2/ev
rm -rf $O; mkdir -p $O
# ---- 1. HBM traffic of the two HBM roofline kernels (separate FETCH_SIZE / WRITE_SIZE passes)
for k in k1 k4; do
  if [ $k = k1 ]; then scr=benchmarks/k1_pmc.py; kn=dwconv7_ln_fwd_strip; else scr=benchmarks/decode_pmc.py; kn=decode_attn_k; fi
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/${k}_f -o p -- python3 $scr > $O/${k}_f.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/${k}_w -o p -- python3 $scr > $O/${k}_w.log 2>&1 || exit 1
  python3 benchmarks/pmc_summary.py $O/${k}_f $O/${k}_w $kn $O/r02_${k}_pmc.json > /dev/null || exit 1
done
# ---- 2. attention counters (final third-generation kernels)
mkdir -p $O/attn
bash benchmarks/run_pmc.sh $O/attn benchmarks/attn_pmc.py -- \
  PASS "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  PASS "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum" \
  PASS "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES TCC_EA0_RDREQ_sum TCC_REQ_sum" \
  || exit $?
python3 benchmarks/attn_pmc_summary.py $O/r02_attn3_pmc.json --kernels=attn_fwd3,attn_bwd_dq3,attn_bwd_dkv3,attn_delta $O/attn/pass1 $O/attn/pass2 $O/attn/pass3 > /dev/null 2>&1
# ---- 3. the train step alone: kernel trace -> per-shape groups and the step timeline
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/step -o t -- python3 bench.py --no-extras --steps 10 --warmup 3 > $O/step.log 2>&1 || exit 1
python3 benchmarks/trace_groups.py $O/step/t_kernel_trace.csv $O/r02_trainstep_kernel_groups.csv
python3 benchmarks/trace_step.py $O/step/t_kernel_trace.csv > $O/r02_trainstep_timeline.txt
grep '"metric"' $O/step.log > $O/r02_trainstep_under_rocprof.json
# ---- 4. the default bench under the profiler: the bench line and the stats its roofline launch times must agree with
cp $O/r02_k1_pmc.json $O/r02_k4_pmc.json profiles/ 2>/dev/null; [ $(stat -c %s $O/r02_attn3_pmc.json) -gt 100 ] && cp $O/r02_attn3_pmc.json profiles/
timeout -k 10 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/full -o t -- python3 bench.py > $O/full.log 2>&1 || exit 1
grep '"metric"' $O/full.log > $O/r02_bench_default_under_rocprof.json
cp $O/full/t_kernel_stats.csv $O/r02_bench_default_kernel_stats.csv
python3 benchmarks/trace_groups.py $O/full/t_kernel_trace.csv $O/r02_bench_default_kernel_groups.csv
# ---- 5. tables: K1 over all shapes, GEMM choices
( cd benchmarks && timeout -k 10 300 python -u kernel_bench.py --what k1all 2>&1 | grep "^K1" ) > $O/r02_k1_all_shapes.txt || exit 1
( cd benchmarks && timeout -k 10 200 python -u fusion_bench.py 2>&1 | grep -v amdgpu.ids; timeout -k 10 200 python -u tn_bench.py 2>&1 | grep dW; timeout -k 10 200 python -u decode_split.py 2>&1 | grep -E "per step|max_len= 80" ) > $O/r02_gemm_and_decode_tables.txt || exit 1
( cd benchmarks && timeout -k 10 200 python -u kernel_bench.py --what attn 2>&1 | grep kernel ) > $O/r02_attn_shapes.txt
rm -rf $O/step $O/full $O/attn/pass*/*.db $O/*_f $O/*_w
find $O -name "*.db" -delete
ls -la $O
