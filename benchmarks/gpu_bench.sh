#!/bin/bash
set -o pipefail
timeout -k 10 400 python -u bench.py --no-extras --steps 20 --warmup 5 2> /dev/null | cut -c1-120
