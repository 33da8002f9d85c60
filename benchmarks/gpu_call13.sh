#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
cd benchmarks
timeout -k 10 200 python -u encoder_bench.py 32 2>&1 | grep encoder

timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d ../gpurun_out/r2/enc_trace -o t -- python3 encoder_bench.py 32 > ../gpurun_out/r2/enc_trace.log 2>&1
cd ..
python3 benchmarks/trace_groups.py gpurun_out/r2/enc_trace/t_kernel_trace.csv gpurun_out/r2/enc_groups.csv
rm -rf gpurun_out/r2/enc_trace
