#!/usr/bin/env python3
"""rope_inplace at the config shape (q|k of [32, 315, 3, 1600], 32 heads of 100, 50 rotated) and the greedy prefill shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops  # noqa: E402
from kernel_bench import timeit  # noqa: E402
dev = torch.device("cuda:0")
for (B, S, H, d) in ((32, 315, 16, 100), (1024, 159, 8, 128)):
    E = H * d
    rot = d // 2
    qkv = (torch.rand(B, S, 3 * E, device=dev) - 0.5).to(torch.bfloat16)
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, device=dev).float() / rot))
    cos_t, sin_t = ops.rope_tables(inv, S)
    ref = qkv.clone()
    ops.rope_inplace(qkv[:, :, :2 * E], 2 * H, rot, cos_t, sin_t)
    ops.rope_inplace(qkv[:, :, :2 * E], 2 * H, rot, cos_t, sin_t, inverse=True)
    print("round trip max err", float((qkv.float() - ref.float()).abs().max()))
    t = timeit(lambda: ops.rope_inplace(qkv[:, :, :2 * E], 2 * H, rot, cos_t, sin_t), iters=50)
    print(f"B{B} S{S} H{H} d{d}: {t * 1e6:.1f} us", flush=True)
