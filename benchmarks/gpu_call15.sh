#!/bin/bash
T="python -u -m pytest tests/test_config_gpu.py -q -s -k test_backward_bf16_model_level_vs_reference_golden --timeout 300 --timeout-method thread"
echo "== current"; MM_TEST_REPORT=1 timeout -k 10 300 $T 2>&1 | grep -E "worst key" | cut -c1-220
echo "== old lib (r10: old K1 + old gemm_big)"; MM_LIB=$PWD/multimodal_amd/lib/libmmhip_r10.so MM_TEST_REPORT=1 timeout -k 10 300 $T 2>&1 | grep -E "worst key" | cut -c1-220
echo "== current, new gemm shapes off"; MM_GEMM_BIG_MIN_MN=4194304 MM_TEST_REPORT=1 timeout -k 10 300 $T 2>&1 | grep -E "worst key" | cut -c1-220
