#!/bin/bash
set -o pipefail
cd benchmarks
for st in 0 1; do for sh in 0 1 2 3; do
echo "== stagger $st shape $sh"
MM_GEMM_BIG_STAGGER=$st MM_GEMM_BIG_SHAPE=$sh timeout -k 10 200 python -u kernel_bench.py --what gemmx --shapes "10080,1600,6400;10080,1600,1600;10080,1600,4800;10080,6400,1600;10080,4800,1600" 2>&1 | grep kernel | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['M'],d['N'],d['K'],'plain %.1f us %.0f TF | bias+res %.1f | lib %.0f TF'%(d['us'],d['TFLOPs'],d['us_bias_res'],d['torch_TFLOPs']))"
done; done
