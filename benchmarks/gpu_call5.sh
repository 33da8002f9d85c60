#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 900 python -u -m pytest tests/test_canary_gpu.py tests/test_edge_gpu.py tests/test_model_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r2/canary.log 2>&1; rc=$?; tail -12 gpurun_out/r2/canary.log; [ $rc -eq 0 ] || exit $rc
for v in "" _r10 _r12; do
  echo "== lib$v"
  MM_LIB=$PWD/multimodal_amd/lib/libmmhip$v.so timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q -k "dwconv or k1 or K1" --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
  MM_LIB=$PWD/multimodal_amd/lib/libmmhip$v.so timeout -k 10 300 python -u benchmarks/kernel_bench.py --what k1all 2>&1 | grep "^K1" | tee gpurun_out/r2/k1all$v.log || exit 1
done
