#!/bin/bash
set -o pipefail
cd benchmarks
timeout -k 10 300 python -u decode_split.py 2>&1 | grep -E "max_len|per step"
