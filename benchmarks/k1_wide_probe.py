#!/usr/bin/env python3
"""Probe: wide K1 (C = 1600) forward/backward at the same byte count with short (L = 31) and long (L = 248) samples."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multimodal_amd import ops
dev = torch.device("cuda:0")
for dtype in (torch.bfloat16, torch.float32):
    for (B, C, Ln) in [(2048, 1600, 31), (256, 1600, 248), (64, 1600, 992)]:
        x = torch.rand(B, Ln, C, device=dev).to(dtype)
        w = torch.randn(C, 1, 7, device=dev) * 0.3
        b = torch.zeros(C, device=dev); lw = torch.ones(C, device=dev); lb = torch.zeros(C, device=dev)
        _, mean, rstd = ops.dwconv7_ln_fwd(x, w, b, lw, lb, save_stats=True)
        dy = torch.randn_like(x)
        tf = bench._event_time(lambda: ops.dwconv7_ln_fwd(x, w, b, lw, lb), iters=10)
        tb = bench._event_time(lambda: ops.dwconv7_ln_bwd(x, dy, None, w, b, lw, mean, rstd), iters=10)
        nb = B * C * Ln * x.element_size()
        print(dtype, B, C, Ln, f"fwd {tf*1e6:.1f} us {2*nb/tf/1e9:.0f} GB/s   bwd {tb*1e6:.1f} us {3*nb/tb/1e9:.0f} GB/s", flush=True)
