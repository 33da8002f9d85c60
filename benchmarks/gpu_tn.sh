#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py tests/test_canary_gpu.py -m gpu -x -q -k "gemm or canary or twice" --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
for st in 1 0; do echo "== TN stagger $st"; MM_GEMM_TN_STAGGER=$st timeout -k 10 200 python -u benchmarks/tn_bench.py 2>&1 | grep dW || exit 1; done
for sh in 0 1 2; do echo "== shape $sh"; MM_GEMM_TN_SHAPE=$sh timeout -k 10 200 python -u benchmarks/tn_bench.py 2>&1 | grep dW || exit 1; done
