#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_attention_gpu.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5 || exit 1
cd benchmarks
for cfg in "1 2 1" "2 2 1"; do
  set -- $cfg
  echo "== NG=$1 NG_DQ=$2 NG_DKV=$3"
  MM_ATTN_NG=$1 MM_ATTN_NG_DQ=$2 MM_ATTN_NG_DKV=$3 timeout -k 10 200 python -u kernel_bench.py --what attn 2>&1 | grep -v amdgpu.ids | cut -c20-75,140-260 || exit 1
done
timeout -k 10 200 python -u kernel_bench.py --what attnx --shapes "32,16,100,1260,884" 2>&1 | grep -v amdgpu.ids | cut -c20-75,140-260
