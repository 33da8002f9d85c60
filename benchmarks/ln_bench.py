#!/usr/bin/env python3
"""LayerNorm backward variants and the decoder's elementwise passes at the config batch (10080 x 1600 / 6400, bf16), alone on the chip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops  # noqa: E402
from kernel_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
rows, C = 10080, 1600
for dt in (torch.bfloat16,):
    x = torch.randn(rows, C, device=dev).to(dt); dy = torch.randn_like(x); dr = torch.randn_like(x)
    w = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    _, mean, rstd = ops.layernorm_fwd(x, w, b, save_stats=True)
    t0 = timeit(lambda: ops.layernorm_fwd(x, w, b, save_stats=True))
    t1 = timeit(lambda: ops.layernorm_bwd(x, dy, dr, w, mean, rstd))
    t2 = timeit(lambda: ops.layernorm_bwd_sums(x, dy, dr, w, mean, rstd))
    t3 = timeit(lambda: ops.colsum(dy, rows, C))
    h = torch.randn(rows, 4 * C, device=dev).to(dt); da = torch.randn_like(h)
    t4 = timeit(lambda: ops.relu2_fwd(h))
    t5 = timeit(lambda: ops.relu2_bwd_colsum_(h, da))
    es = x.element_size()
    gb = rows * C * es / 1e9
    print(f"LN fwd {t0*1e6:6.1f} us ({2*gb/t0:5.0f} GB/s) | LN bwd+res {t1*1e6:6.1f} us ({4*gb/t1:5.0f} GB/s) | LN bwd+res+sums {t2*1e6:6.1f} us ({4*gb/t2:5.0f} GB/s) "
          f"| colsum {t3*1e6:6.1f} us | relu2 fwd {t4*1e6:6.1f} us ({8*gb/t4:5.0f} GB/s) | relu2 bwd+colsum {t5*1e6:6.1f} us ({12*gb/t5:5.0f} GB/s)", flush=True)
