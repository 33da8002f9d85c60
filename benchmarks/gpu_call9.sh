#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_model_gpu.py tests/test_config_gpu.py tests/test_edge_gpu.py -x -q -k "greedy or generate" --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
cd benchmarks
timeout -k 10 300 python -u decode_bench.py gemm 2>&1 | grep -E "^gemm|tokens_per_s" || exit 1
MM_DECODE_GEMM=lib timeout -k 10 300 python -u decode_bench.py 2>&1 | grep -E "tokens_per_s" || exit 1
