#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
MM_TEST_REPORT=1 timeout -k 10 900 python -u -m pytest tests/test_config_gpu.py tests/test_model_gpu.py -k "bf16" -v -s --timeout 300 --timeout-method thread > gpurun_out/r2/config.log 2>&1; rc=$?
grep -E "FAILED|ERROR|bf16 |assert|Error" gpurun_out/r2/config.log | cut -c1-260 | tail -50
exit $rc
