#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r2; cd benchmarks
timeout -k 10 300 python -u decode_bench.py 2>&1 | grep tokens_per_s || exit 1
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace -d ../gpurun_out/r2/decode_trace -o t -- python3 decode_bench.py > ../gpurun_out/r2/decode_trace.log 2>&1 || exit 1
grep tokens_per_s ../gpurun_out/r2/decode_trace.log
cd ..
python3 benchmarks/trace_busy.py gpurun_out/r2/decode_trace/t_kernel_trace.csv 0.30 0.45
rm -rf gpurun_out/r2/decode_trace
