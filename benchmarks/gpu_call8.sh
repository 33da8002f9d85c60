#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py tests/test_canary_gpu.py -x -q -k "dwconv or k1 or K1 or twice" --timeout 120 --timeout-method thread 2>&1 | tail -12 || exit 1
for v in 0; do
  echo "== wide variant $v"
  MM_K1W_VARIANT=$v timeout -k 10 300 python -u benchmarks/kernel_bench.py --what k1all 2>&1 | grep "^K1" | grep "C1600\|C2048" || exit 1
done
