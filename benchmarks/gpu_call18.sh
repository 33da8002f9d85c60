#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
cd benchmarks && timeout -k 10 200 python -u fusion_bench.py 2>&1 | grep -v amdgpu.ids | tee ../gpurun_out/r2/fusion_bench.log || exit 1
cd ..
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/r2/train_trace -o t -- python3 bench.py --no-extras --steps 10 --warmup 3 > gpurun_out/r2/train_trace.log 2>&1 || exit 1
python3 benchmarks/trace_groups.py gpurun_out/r2/train_trace/t_kernel_trace.csv gpurun_out/r2/train_groups.csv
cp gpurun_out/r2/train_trace/t_kernel_stats.csv gpurun_out/r2/train_kernel_stats.csv
rm -rf gpurun_out/r2/train_trace
