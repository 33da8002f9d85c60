#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_model_gpu.py tests/test_config_gpu.py -x -q -k "own_gemm or backward or bf16" --timeout 180 --timeout-method thread 2>&1 | tail -4 || exit 1
cd benchmarks; timeout -k 10 200 python -u tn_bench.py 2>&1 | grep "dW"; cd ..
echo "auto:"; timeout -k 10 400 python -u bench.py --no-extras --steps 20 --warmup 5 2> /dev/null | cut -c30-130
echo "own:"; MM_GEMM_BACKEND=own timeout -k 10 400 python -u bench.py --no-extras --steps 20 --warmup 5 2>/dev/null | cut -c30-130
