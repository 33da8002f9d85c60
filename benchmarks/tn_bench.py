#!/usr/bin/env python3
"""Weight-gradient products of a decoder layer (dW = dY^T X, float32 result) at the config batch: the library, gemm_k's
128x128 transposed-read tiles, and the large-tile weight-gradient kernel (per tile shape / K split via MM_GEMM_TN_*)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd  # noqa: E402,F401
from multimodal_amd import ops  # noqa: E402
from kernel_bench import timeit  # noqa: E402

d = torch.device("cuda:0")
M, E, F4 = 10080, 1600, 6400
r = lambda *s: (torch.rand(*s, device=d) * 2 - 1).to(torch.bfloat16)
for (n, k, nm) in ((F4, E, "dW1"), (E, F4, "dW2"), (3 * E, E, "dWqkv"), (E, E, "dWo")):
    dy, x = r(M, n), r(M, k)
    out = torch.empty(n, k, device=d)
    t_lib = timeit(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32, out=out))
    t_own = timeit(lambda: ops.gemm(dy, x, n, k, M, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32, out=out))
    fl = 2.0 * M * n * k
    print(f"{nm}: lib {t_lib*1e6:.1f} us ({fl/t_lib/1e15:.2f} PF) | libmmhip {t_own*1e6:.1f} us ({fl/t_own/1e15:.2f} PF)", flush=True)
