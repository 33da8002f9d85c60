#!/bin/bash
# large-tile GEMM main-loop A/B on the decoder shapes: parity tests, then staggered (late DMA issue) vs lockstep vs the library,
# forward form and weight-gradient form
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" --timeout 120 --timeout-method thread 2>&1 | tail -4 || exit 1
S="10080,1600,6400;10080,6400,1600;10080,4800,1600;10080,1600,1600;10080,640,1600;992,6400,1600;8192,8192,8192"
for mode in 1 0; do
echo "== stagger $mode"; MM_GEMM_BIG_STAGGER=$mode timeout -k 10 200 python -u benchmarks/kernel_bench.py --what gemmx --shapes "$S" 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['M'], r['N'], r['K'], '%.1f us %.3f PF | lib %.3f PF | fused bias+res %.1f us lib %.1f us' % (r['us'], r['TFLOPs']/1e3, r['torch_TFLOPs']/1e3, r.get('us_bias_res', 0), r.get('torch_us_bias_res', 0)))
" || exit 1
MM_GEMM_BIG_STAGGER=$mode timeout -k 10 200 python -u benchmarks/tn_bench.py || exit 1
done
