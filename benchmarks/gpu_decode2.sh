#!/bin/bash
set -o pipefail
cd benchmarks
timeout -k 10 300 python -u decode_bench.py gemm 2>&1 | grep -E "gemm M|tokens_per_s"
