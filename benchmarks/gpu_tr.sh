#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_transpose_multi_gpu.py tests/test_optim_gpu.py tests/test_graph_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5 || exit 1
VARIANTS="MM_TRANSPOSE_MULTI=0" bash benchmarks/gpu_step_ab.sh
