#!/bin/bash
set -o pipefail
cd benchmarks
timeout -k 10 300 python -u dw_mode_bench.py 2>&1 | grep "DP="
