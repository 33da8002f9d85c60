#!/usr/bin/env python3
"""Weight-gradient products of a decoder layer at the config batch (K = B*S = 10080 tokens, both operands token-major,
float32 result): dW[out,in] = dY[tok,out]^T . X[tok,in] -- libmmhip (gemm_k, transposing LDS reads) vs the library."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd  # noqa: F401,E402  (sets the library's data-parallel mode)
from multimodal_amd import ops  # noqa: E402
from kernel_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10080
for (Mo, No) in [(6400, 1600), (1600, 6400), (4800, 1600), (1600, 1600)]:
    dy = (torch.rand(K, Mo, device=dev) - 0.5).to(torch.bfloat16)
    x = (torch.rand(K, No, device=dev) - 0.5).to(torch.bfloat16)
    out = torch.empty(Mo, No, device=dev, dtype=torch.float32)
    t1 = timeit(lambda: ops.gemm(dy, x, Mo, No, K, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, out=out), iters=10)
    ref = torch.mm(dy.t(), x, out_dtype=torch.float32)
    err = float((out - ref).abs().max() / ref.abs().max())
    t2 = timeit(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32, out=out), iters=10)
    fl = 2.0 * Mo * No * K
    print(f"dW {Mo}x{No} K{K}: libmmhip {t1 * 1e6:7.1f} us {fl / t1 / 1e12:6.0f} TF/s | library {t2 * 1e6:7.1f} us {fl / t2 / 1e12:6.0f} TF/s | rel diff {err:.1e}",
          flush=True)
