#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace -d gpurun_out/r2/tt -o t -- python3 bench.py --no-extras --steps 6 --warmup 3 > gpurun_out/r2/tt.log 2>&1 || exit 1
head -1 gpurun_out/r2/tt/t_kernel_trace.csv
python3 benchmarks/trace_step.py gpurun_out/r2/tt/t_kernel_trace.csv
rm -rf gpurun_out/r2/tt
