#!/bin/bash
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -6
