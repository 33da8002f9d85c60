#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" --timeout 120 --timeout-method thread 2>&1 | tail -12 || exit 1
cd benchmarks
echo "== default"; timeout -k 10 200 python -u tn_bench.py 2>&1 | grep "dW"
echo "== gemm_k tiles (MM_GEMM_NO_TN)"; MM_GEMM_NO_TN=1 timeout -k 10 200 python -u tn_bench.py 2>&1 | grep "dW"
for sh in 0 1 2; do for sp in 1 2 3 4; do echo "== shape $sh split $sp"; MM_GEMM_TN_SHAPE=$sh MM_GEMM_TN_SPLITS=$sp timeout -k 10 200 python -u tn_bench.py 2>&1 | grep "dW" | cut -d'|' -f2; done; done
