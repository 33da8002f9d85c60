#!/usr/bin/env python3
"""K1b strip kernels at the three encoder-stage shapes of bench.py's all-shape table, for one strip-length cap
(MM_K1B_TLMAX, read once per process): time and fraction of the 8 TB/s peak."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multimodal_amd import ops
dev = torch.device("cuda:0")
for dtype in (torch.bfloat16, torch.float32):
    for (B, C, Ln) in [(8192, 64, 250), (8192, 128, 125), (8192, 256, 62)]:
        x = torch.rand(B, Ln, C, device=dev).to(dtype)
        w = torch.randn(C, 1, 7, device=dev) * 0.3
        b = torch.zeros(C, device=dev); lw = torch.ones(C, device=dev); lb = torch.zeros(C, device=dev)
        _, mean, rstd = ops.dwconv7_ln_fwd(x, w, b, lw, lb, save_stats=True)
        dy = torch.randn_like(x)
        tb = bench._event_time(lambda: ops.dwconv7_ln_bwd(x, dy, None, w, b, lw, mean, rstd), iters=20)
        nb = B * C * Ln * x.element_size()
        print(f"TLMAX={os.environ.get('MM_K1B_TLMAX', '64'):>4s} {str(dtype):15s} B{B} C{C} L{Ln}: bwd {tb*1e6:7.1f} us {3*nb/tb/1e9:6.0f} GB/s ({3*nb/tb/8e12:.3f})", flush=True)
