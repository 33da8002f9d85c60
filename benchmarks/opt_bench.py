#!/usr/bin/env python3
"""OrthoGrad-AdamW step alone on the config model's parameter set (f32 masters + bf16 operand copies): time and HBM rate
(34 bytes per parameter in the update pass, 8 in the reduce pass)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd  # noqa: E402,F401
from multimodal_amd.optim import OrthoGradAdamW  # noqa: E402
from kernel_bench import timeit  # noqa: E402

d = torch.device("cuda:0")
E, F4, L = 1600, 6400, 8
shapes = []
for _ in range(L):
    shapes += [(E, E)] * 4 + [(F4, E), (E, F4)] + [(E,)] * 9 + [(F4,)]
shapes += [(600, E), (E, 600)]
ps = [torch.nn.Parameter(torch.randn(*s, device=d) * 0.02) for s in shapes]
for p in ps:
    p.grad = torch.randn_like(p) * 0.01
opt = OrthoGradAdamW(ps, lr=1e-4, weight_decay=0.01)
n = sum(p.numel() for p in ps)
t = timeit(lambda: opt.step(), iters=20, warmup=3)
print(f"{n/1e6:.1f} M parameters: step {t*1e3:.3f} ms = {n*42/t/1e12:.2f} TB/s over both passes (42 B per parameter; bf16 copies only when registered)")
