#!/usr/bin/env python3
"""Only the large-tile GEMM launches of bench.py's MFMA roofline leg (10080 x 1600 x 6400, bias + residual), for rocprofv3 --pmc."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (10080, 1600, 6400)
dev = torch.device("cuda:0")
a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
w = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
bias = torch.zeros(N, device=dev)
res = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(9):
    ops.gemm(a, w, M, N, K, bias=bias, res=res)
torch.cuda.synchronize()
