#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_canary_gpu.py -x -q -k "dwconv or k1 or K1 or canary or determin" --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
cd benchmarks && timeout -k 10 300 python -u kernel_bench.py --what k1all 2>&1 | grep "K1 bwd"
