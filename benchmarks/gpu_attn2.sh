#!/bin/bash
set -o pipefail
cd benchmarks
timeout -k 10 200 python -u kernel_bench.py --what attnx --shapes "32,16,100,158,110;32,16,100,315,221;32,16,100,630,442;32,16,100,1260,884;32,16,100,315,315;32,16,128,315,221;8,16,100,315,221" 2>&1 | grep -v amdgpu.ids | cut -c20-75,140-260
