#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_model_gpu.py -x -q -k "own_gemm" --timeout 180 --timeout-method thread 2>&1 | tail -15 || exit 1
echo "auto:"; timeout -k 10 400 python -u bench.py --no-extras --steps 20 --warmup 5 2> /dev/null | cut -c30-130
echo "own:"; MM_GEMM_BACKEND=own timeout -k 10 400 python -u bench.py --no-extras --steps 20 --warmup 5 2>&1 | tail -3 | cut -c1-200
