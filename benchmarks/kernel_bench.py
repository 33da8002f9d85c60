#!/usr/bin/env python3
"""Per-kernel microbenchmarks (GPU): achieved GB/s or TFLOP/s from algorithmic bytes/flops."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops, _lib as L  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def bench_k1(B, C, Ln, dtype, bwd=False):
    d = torch.device("cuda:0")
    x = torch.rand(B, Ln, C, device=d).to(dtype)
    w = torch.randn(C, 1, 7, device=d); b = torch.zeros(C, device=d); lw = torch.ones(C, device=d); lb = torch.zeros(C, device=d)
    es = x.element_size()
    if not bwd:
        t = timeit(lambda: ops.dwconv7_ln_fwd(x, w, b, lw, lb))
        nbytes = B * C * Ln * 2 * es
    else:
        y, mean, rstd = ops.dwconv7_ln_fwd(x, w, b, lw, lb, save_stats=True)
        dy = torch.randn_like(x)
        t = timeit(lambda: ops.dwconv7_ln_bwd(x, dy, None, w, b, lw, mean, rstd))
        nbytes = B * C * Ln * 3 * es
    return dict(kernel="dwconv7_ln_" + ("bwd" if bwd else "fwd"), B=B, C=C, L=Ln, dtype=str(dtype), us=t * 1e6, GBps=nbytes / t / 1e9)


def bench_gemm(M, N, K, dtype, akm=False, bkm=False):
    d = torch.device("cuda:0")
    a = (torch.rand((K, M) if akm else (M, K), device=d) * 2 - 1).to(dtype)
    b = (torch.rand((K, N) if bkm else (N, K), device=d) * 2 - 1).to(dtype)
    t = timeit(lambda: ops.gemm(a, b, M, N, K, a_kmajor=akm, b_kmajor=bkm))
    tt = timeit(lambda: torch.matmul(a.t() if akm else a, b if bkm else b.t()))
    r = dict(kernel="gemm", M=M, N=N, K=K, dtype=str(dtype), layout=f"{int(akm)}{int(bkm)}", us=t * 1e6,
             TFLOPs=2 * M * N * K / t / 1e12, torch_TFLOPs=2 * M * N * K / tt / 1e12)
    if not akm and not bkm:            # fused bias + residual epilogue vs library addmm + add
        bias = torch.randn(N, device=d); res = torch.randn(M, N, device=d).to(dtype)
        te = timeit(lambda: ops.gemm(a, b, M, N, K, bias=bias, res=res))
        bb = bias.to(dtype)
        tte = timeit(lambda: torch.addmm(bb, a, b.t()).add_(res))
        r.update(us_bias_res=te * 1e6, torch_us_bias_res=tte * 1e6)
    return r


def bench_attn(B, H, d, S, P, dtype, stablemax=True):
    dev = torch.device("cuda:0")
    E = H * d
    rot = int(d * 0.5); rot -= rot % 2
    qkv = (torch.rand(B, S, 3, E, device=dev) * 2 - 1).to(dtype)
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, device=dev).float() / rot))
    cos_t, sin_t = ops.rope_tables(inv, S)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    ops.rope_inplace(q, H, rot, cos_t, sin_t); ops.rope_inplace(k, H, rot, cos_t, sin_t)   # as the model does
    o, lse = ops.prefix_attn_fwd(q, k, v, H, 0, P, cos_t, sin_t, stablemax, save_lse=True)
    do = torch.randn_like(o)
    tf = timeit(lambda: ops.prefix_attn_fwd(q, k, v, H, 0, P, cos_t, sin_t, stablemax, save_lse=True))
    tb = timeit(lambda: ops.prefix_attn_bwd(q, k, v, o, do, lse, H, 0, P, cos_t, sin_t, stablemax))
    fl = 4.0 * B * S * S * E
    return dict(kernel="prefix_attn", B=B, H=H, d=d, S=S, P=P, dtype=str(dtype), stablemax=stablemax, fwd_us=tf * 1e6,
                bwd_us=tb * 1e6, fwd_TFLOPs=fl / tf / 1e12, bwd_TFLOPs=2.5 * fl / tb / 1e12)


def bench_ln(rows, C, dtype):
    d = torch.device("cuda:0")
    x = torch.randn(rows, C, device=d).to(dtype)
    w = torch.ones(C, device=d); b = torch.zeros(C, device=d)
    y, mean, rstd = ops.layernorm_fwd(x, w, b, save_stats=True)
    dy = torch.randn_like(x); dres = torch.randn_like(x)
    tf = timeit(lambda: ops.layernorm_fwd(x, w, b, save_stats=True))
    tb = timeit(lambda: ops.layernorm_bwd(x, dy, dres, w, mean, rstd))
    es = x.element_size()
    return dict(kernel="layernorm", rows=rows, C=C, dtype=str(dtype), fwd_us=tf * 1e6, bwd_us=tb * 1e6,
                fwd_GBps=rows * C * 2 * es / tf / 1e9, bwd_GBps=rows * C * 4 * es / tb / 1e9)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="k1,gemm")
    ap.add_argument("--shapes", default="", help="gemm shapes M,N,K;M,N,K... (bf16 NT) for --what gemmx")
    a = ap.parse_args()
    res = []
    if "quick1" in a.what:
        for dt in (torch.float32, torch.bfloat16):
            for (B, C, Ln) in [(8192, 64, 250), (8192, 256, 62), (2048, 1600, 31)]:
                res.append(bench_k1(B, C, Ln, dt)); print(json.dumps(res[-1]), flush=True)
            res.append(bench_k1(8192, 64, 250, dt, bwd=True)); print(json.dumps(res[-1]), flush=True)
    if "k1all" in a.what.split(","):     # the four stage shapes x dtype x fwd/bwd at a saturating batch (compact lines)
        for dt in (torch.float32, torch.bfloat16):
            for (B, C, Ln) in [(8192, 64, 250), (8192, 128, 125), (8192, 256, 62), (2048, 1600, 31), (2048, 2048, 31)]:
                for bwd in (False, True):
                    r = bench_k1(B, C, Ln, dt, bwd=bwd)
                    print(f"K1 {'bwd' if bwd else 'fwd'} {str(dt)[6:]:8s} B{B} C{C} L{Ln}: {r['us']:8.1f} us {r['GBps']:7.0f} GB/s "
                          f"({r['GBps'] / 8000:.3f} of 8 TB/s)", flush=True)
    if "k1" in a.what.split(","):
        for dt in (torch.float32, torch.bfloat16):
            for (B, C, Ln) in [(8192, 64, 250), (8192, 128, 125), (8192, 256, 62), (4096, 1600, 31), (32, 64, 250), (32, 256, 62), (32, 1600, 31)]:
                res.append(bench_k1(B, C, Ln, dt)); print(json.dumps(res[-1]), flush=True)
            for (B, C, Ln) in [(8192, 64, 250), (8192, 256, 62), (2048, 1600, 31), (32, 64, 250)]:
                res.append(bench_k1(B, C, Ln, dt, bwd=True)); print(json.dumps(res[-1]), flush=True)
    if "gemm" in a.what.split(","):
        for dt in (torch.bfloat16, torch.float32):
            for (M, N, K) in [(10080, 6400, 1600), (10080, 1600, 6400), (10080, 1600, 1600), (992, 6400, 1600), (992, 1600, 6400), (8000, 256, 64), (4096, 4096, 4096)]:
                res.append(bench_gemm(M, N, K, dt)); print(json.dumps(res[-1]), flush=True)
            res.append(bench_gemm(10080, 1600, 6400, dt, False, True)); print(json.dumps(res[-1]), flush=True)
            res.append(bench_gemm(6400, 1600, 10080, dt, True, True)); print(json.dumps(res[-1]), flush=True)
    if "ln" in a.what.split(","):
        for dt in (torch.bfloat16, torch.float32):
            for (rows, C) in [(10080, 1600), (80640, 1600), (8000, 64), (992, 1600)]:
                res.append(bench_ln(rows, C, dt)); print(json.dumps(res[-1]), flush=True)
    if "gemmx" in a.what:
        for sh in a.shapes.split(";"):
            M, N, K = (int(v) for v in sh.split(","))
            res.append(bench_gemm(M, N, K, torch.bfloat16)); print(json.dumps(res[-1]), flush=True)
    if "attnx" in a.what:
        for sh in a.shapes.split(";"):
            B, H, d, S, P = (int(v) for v in sh.split(","))
            res.append(bench_attn(B, H, d, S, P, torch.bfloat16, True)); print(json.dumps(res[-1]), flush=True)
    if "attn" in a.what.split(","):
        for (B, H, d, S, P, sm) in [(32, 16, 100, 315, 221, True), (32, 8, 128, 238, 159, False), (32, 4, 192, 315, 221, True), (32, 4, 512, 315, 221, True)]:
            res.append(bench_attn(B, H, d, S, P, torch.bfloat16, sm)); print(json.dumps(res[-1]), flush=True)
