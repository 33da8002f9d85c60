#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 900 python -u bench.py > gpurun_out/r2/bench_full.json 2> gpurun_out/r2/bench_full.err
echo rc=$?
tail -3 gpurun_out/r2/bench_full.err
