#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" --timeout 120 --timeout-method thread 2>&1 | tail -1 || exit 1
timeout -k 10 400 python -u bench.py --no-extras --steps 20 --warmup 5 2> /dev/null | cut -c1-200
cd benchmarks && timeout -k 10 200 python -u encoder_bench.py 32 2>&1 | grep encoder
