#!/bin/bash
# copy the summaries of the last scratch/deliver.sh run into profiles/ (newest trace only)
set -e
f=$(ls -t gpurun_out/r01/bench/vm/*_kernel_stats.csv | head -1)
t=${f/_kernel_stats/_kernel_trace}
cp $f profiles/r01_bench_default_kernel_stats.csv
python benchmarks/trace_groups.py $t profiles/r01_bench_default_kernel_groups.csv
cp gpurun_out/r01/bench_default.json profiles/r01_bench_default_under_rocprof.json
cp gpurun_out/r01/k1_pmc.json profiles/r01_k1_pmc.json
cp gpurun_out/r01/k4_pmc.json profiles/r01_k4_pmc.json
ls -t gpurun_out/r01/bench/vm/*_kernel_stats.csv | tail -n +2 | sed 's/_kernel_stats.csv//' | while read p; do rm -f ${p}_*; done
