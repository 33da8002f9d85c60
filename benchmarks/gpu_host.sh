#!/bin/bash
set -o pipefail
cd benchmarks
timeout -k 10 300 python -u step_host.py 2>&1 | grep -E "per step|per phase|drained|ms from"
