#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 900 python -u -m pytest tests/test_model_gpu.py tests/test_config_gpu.py tests/test_edge_gpu.py tests/test_graph_gpu.py tests/test_dp_gpu.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 400 python -u bench.py --no-extras --steps 20 --warmup 5 2> gpurun_out/r2/bench_train.err | tee gpurun_out/r2/bench_train.json | cut -c1-200
cd benchmarks && timeout -k 10 300 python -u decode_bench.py 2>&1 | grep -E "tokens_per_s"
