#!/bin/bash
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py tests/test_canary_gpu.py -m gpu -x -q -k "gemm or canary or twice" --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
echo "== auto shape"; timeout -k 10 200 python -u benchmarks/tn_bench.py 2>&1 | grep dW || exit 1
for sh in 0 1 2; do echo "== shape $sh"; MM_GEMM_TN_SHAPE=$sh timeout -k 10 200 python -u benchmarks/tn_bench.py 2>&1 | grep dW || exit 1; done
S="10080,1600,6400;10080,6400,1600;10080,4800,1600;10080,1600,1600;10080,640,1600;992,6400,1600;8192,8192,8192"
timeout -k 10 200 python -u benchmarks/kernel_bench.py --what gemmx --shapes "$S" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['M'], r['N'], r['K'], '%.1f us %.3f PF | lib %.3f PF | fused bias+res %.1f us lib %.1f us' % (r['us'], r['TFLOPs']/1e3, r['torch_TFLOPs']/1e3, r.get('us_bias_res', 0), r.get('torch_us_bias_res', 0)))
"
