#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/all.log 2>&1; rc=$?; tail -3 gpurun_out/r2/all.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python -u bench.py --no-extras --steps 20 --warmup 5 2> gpurun_out/r2/bench_train.err | cut -c1-200
cd benchmarks && timeout -k 10 200 python -u encoder_bench.py 32 2>&1 | grep encoder
