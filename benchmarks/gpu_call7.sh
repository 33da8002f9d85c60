#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py tests/test_canary_gpu.py -x -q -k "dwconv or k1 or K1 or twice" --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
for qd in 2 4 8; do
  echo "== fwd QD=$qd, bwd QD=$((qd>4?4:qd))"
  MM_K1_QD=$qd MM_K1B_QD=$((qd>4?4:qd)) timeout -k 10 300 python -u benchmarks/kernel_bench.py --what k1all 2>&1 | grep "^K1" | grep -v "C1600\|C2048" || exit 1
done
