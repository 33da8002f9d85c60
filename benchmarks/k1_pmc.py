#!/usr/bin/env python3
"""Runs only the K1 forward launches of bench.py's roofline leg (same shape), for rocprofv3 --pmc passes:
    rocprofv3 --pmc FETCH_SIZE -d out_f -- python3 benchmarks/k1_pmc.py
    rocprofv3 --pmc WRITE_SIZE -d out_w -- python3 benchmarks/k1_pmc.py
benchmarks/pmc_summary.py turns the two counter CSVs into profiles/*_k1_pmc.json (per-launch HBM bytes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops  # noqa: E402

dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == "bf16" else torch.float32
bwd = len(sys.argv) > 2 and sys.argv[2] == "bwd"
B, C, Ln = 8192, 64, 250
if len(sys.argv) > 5:
    B, C, Ln = (int(v) for v in sys.argv[3:6])
d = torch.device("cuda:0")
x = torch.rand(B, Ln, C, device=d).to(dt)
w = torch.randn(C, 1, 7, device=d) * 0.3
b = torch.zeros(C, device=d); lw = torch.ones(C, device=d); lb = torch.zeros(C, device=d)
if bwd:
    y, mean, rstd = ops.dwconv7_ln_fwd(x, w, b, lw, lb, save_stats=True)
    dy = torch.randn_like(x)
for _ in range(8):
    if bwd:
        ops.dwconv7_ln_bwd(x, dy, None, w, b, lw, mean, rstd)
    else:
        ops.dwconv7_ln_fwd(x, w, b, lw, lb)
torch.cuda.synchronize()
