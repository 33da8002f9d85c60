#!/bin/bash
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_model_gpu.py tests/test_config_gpu.py tests/test_attention_gpu.py -x -q -k "greedy or generate or kv or decode" --timeout 180 --timeout-method thread 2>&1 | tail -4 || exit 1
cd benchmarks
timeout -k 10 300 python -u decode_bench.py 2>&1 | grep -E "tokens_per_s"
