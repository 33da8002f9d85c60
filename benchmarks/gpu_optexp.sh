#!/bin/bash
set -o pipefail
cd benchmarks
timeout -k 10 300 python -u opt_overlap_exp.py 1 2>&1 | grep -E "plain|overlap|rror" || exit 1
timeout -k 10 300 python -u opt_overlap_exp.py 2 2>&1 | grep -E "plain|overlap|rror"
