#!/bin/bash
# Round-3 evidence: everything profiles/r03_* holds comes from this one script (one gpurun call) + benchmarks/gpu_evidence_attn3.sh.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3/ev
mkdir -p $O
# ---- 1. HBM traffic of the two HBM roofline kernels (separate FETCH_SIZE / WRITE_SIZE passes)
for k in k1 k4; do
  if [ $k = k1 ]; then scr=benchmarks/k1_pmc.py; kn=dwconv7_ln_fwd_strip; else scr=benchmarks/decode_pmc.py; kn=decode_attn1_k; fi
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/${k}_f -o p -- python3 $scr > $O/${k}_f.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/${k}_w -o p -- python3 $scr > $O/${k}_w.log 2>&1 || exit 1
  python3 benchmarks/pmc_summary.py $O/${k}_f $O/${k}_w $kn $O/r03_${k}_pmc.json > /dev/null || exit 1
done
cp $O/r03_k1_pmc.json $O/r03_k4_pmc.json profiles/ 2>/dev/null
# ---- 2. the train step alone: kernel trace -> per-shape groups and the step timeline
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/step -o t -- python3 bench.py --no-extras --steps 10 --warmup 3 > $O/step.log 2>&1 || exit 1
python3 benchmarks/trace_groups.py $O/step/t_kernel_trace.csv $O/r03_trainstep_kernel_groups.csv
python3 benchmarks/trace_step.py $O/step/t_kernel_trace.csv > $O/r03_trainstep_timeline.txt
grep '"metric"' $O/step.log > $O/r03_trainstep_under_rocprof.json
# ---- 3. the default bench under the profiler: the bench line and the stats its roofline launch times must agree with
timeout -k 10 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/full -o t -- python3 bench.py > $O/full.log 2>&1 || exit 1
grep '"metric"' $O/full.log > $O/r03_bench_default_under_rocprof.json
cp $O/full/t_kernel_stats.csv $O/r03_bench_default_kernel_stats.csv
python3 benchmarks/trace_groups.py $O/full/t_kernel_trace.csv $O/r03_bench_default_kernel_groups.csv
# ---- 4. tables
( cd benchmarks && timeout -k 10 300 python -u kernel_bench.py --what k1all 2>&1 | grep "^K1" ) > $O/r03_k1_all_shapes.txt || exit 1
( cd benchmarks && timeout -k 10 200 python -u kernel_bench.py --what attn 2>&1 | grep kernel ) > $O/r03_attn_shapes.txt
( for s in 2 1; do echo "MM_DECODE_SPLIT=$s"; MM_DECODE_SPLIT=$s timeout -k 10 200 python benchmarks/greedy_ab.py 2>&1 | grep -E "tok/s|K4"; done ) > $O/r03_greedy_split_ab.txt
timeout -k 10 300 python benchmarks/step_gaps.py 2>/dev/null | grep -v amdgpu > $O/r03_step_gaps_unprofiled.json
rm -rf $O/step $O/full $O/*_f $O/*_w
find $O -name "*.db" -delete
ls -la $O
