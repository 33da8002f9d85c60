#!/usr/bin/env python3
"""The ConvNeXt encoder's two pointwise products per stage at a given batch (BATCH, default 1024; stage widths of the nocat
config), inference form: time, the bytes they must move (read A, write out, + residual) and the fraction of the 8 TB/s peak.
MM_GEMM_NO256=1 sends them to the small-tile kernel; LIB=1 times torch (library product + separate GELU / scale-add)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multimodal_amd import ops
dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", "1024"))
lib = os.environ.get("LIB") == "1"
stages = [(64, 250), (128, 125), (256, 62), (512, 31)]
if os.environ.get("STAGES"):                       # e.g. STAGES=1600:31 (the headline config's last stage)
    stages = [tuple(int(v) for v in st.split(":")) for st in os.environ["STAGES"].split(",")]
for (C, Ln) in stages:
    M = B * Ln
    xn = torch.randn(M, C, device=dev).to(torch.bfloat16)
    x = torch.randn(M, C, device=dev).to(torch.bfloat16)
    w1 = (torch.randn(4 * C, C, device=dev) * 0.05).to(torch.bfloat16); b1 = torch.zeros(4 * C, device=dev)
    w2 = (torch.randn(C, 4 * C, device=dev) * 0.05).to(torch.bfloat16); b2 = torch.zeros(C, device=dev)
    g = torch.ones(C, device=dev)
    if lib:
        b1h, b2h = b1.to(torch.bfloat16), b2.to(torch.bfloat16)
        f1 = lambda: torch.nn.functional.gelu(torch.addmm(b1h, xn, w1.t()))
        a = f1()
        f2 = lambda: torch.addmm(b2h, a, w2.t()).mul_(g.to(torch.bfloat16)).add_(x)
    else:
        f1 = lambda: ops.gemm(xn, w1, M, 4 * C, C, bias=b1, act=ops.ACT_GELU)
        a = f1()
        f2 = lambda: ops.gemm(a, w2, M, C, 4 * C, bias=b2, gamma=g, res=x)
    t1 = bench._event_time(f1, iters=10)
    t2 = bench._event_time(f2, iters=10)
    by1 = M * C * 2 + M * 4 * C * 2
    by2 = M * 4 * C * 2 + 2 * M * C * 2
    print(f"B{B} C{C:4d} L{Ln:4d} rows {M:7d}: pwconv1+GELU {t1*1e6:7.1f} us ({by1/t1/8e12:.2f} of HBM, {2*M*C*4*C/t1/1e12:6.1f} TF/s)   "
          f"pwconv2+scale+res {t2*1e6:7.1f} us ({by2/t2/8e12:.2f} of HBM, {2*M*C*4*C/t2/1e12:6.1f} TF/s)", flush=True)
