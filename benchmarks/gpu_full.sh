#!/bin/bash
# full GPU suite + the K1 table (used after kernel changes)
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/all.log 2>&1; rc=$?; tail -4 gpurun_out/r2/all.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u benchmarks/kernel_bench.py --what k1all 2>&1 | grep "^K1" | tee gpurun_out/r2/k1all_final.log
