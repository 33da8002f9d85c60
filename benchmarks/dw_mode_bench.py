#!/usr/bin/env python3
"""Weight-gradient products of a decoder layer at the config batch (dW = dY^T X, float32 result) on the library, in its
data-parallel tile mode (the package default) against its stream-K mode (TENSILE_STREAMK_DATA_PARALLEL unset/0)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd  # noqa: E402,F401
from kernel_bench import timeit  # noqa: E402

d = torch.device("cuda:0")
M, E, F4 = 10080, 1600, 6400
r = lambda *s: (torch.rand(*s, device=d) * 2 - 1).to(torch.bfloat16)
for mode in ("1", "0", "1", "0"):
    os.environ["TENSILE_STREAMK_DATA_PARALLEL"] = mode
    for (n, k, nm) in ((F4, E, "dW1"), (E, F4, "dW2"), (3 * E, E, "dWqkv"), (E, E, "dWo")):
        dy, x = r(M, n), r(M, k)
        out = torch.empty(n, k, device=d)
        t = timeit(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32, out=out))
        print(f"DP={mode} {nm}: {t*1e6:.1f} us = {2*M*n*k/t/1e15:.2f} PF", flush=True)
