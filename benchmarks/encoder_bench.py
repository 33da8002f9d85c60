#!/usr/bin/env python3
"""The IR encoder alone (forward + backward, bf16, E = 1600) at the config batch and a saturating batch: wall time per
pass; under rocprofv3 --kernel-trace the per-kernel split (no decoder work overlapping it)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd as mm  # noqa: E402
from multimodal_amd import functional as FN  # noqa: E402
from multimodal_amd.models import MultimodalSpectralEncoder  # noqa: E402

dev = torch.device("cuda:0")
E = int(sys.argv[2]) if len(sys.argv) > 2 else 1600
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mm.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
enc = MultimodalSpectralEncoder(embed_dim=E, verbose=False).to(dev).train()
ir = torch.rand(B, 1000, device=dev)
g = None


def step():
    global g
    for p in enc.parameters():
        p.grad = None
    mem = enc(None, ir, None)
    if g is None:
        g = torch.randn_like(mem)
    mem.backward(g)
    FN.join_side_stream()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    step()
torch.cuda.synchronize()
print(f"encoder fwd+bwd B={B} E={E}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per pass", flush=True)
with torch.no_grad():
    enc.eval()
    for _ in range(3):
        enc(None, ir, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        enc(None, ir, None)
    torch.cuda.synchronize()
    print(f"encoder fwd (eval) B={B} E={E}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms", flush=True)
