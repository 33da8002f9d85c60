#!/usr/bin/env python3
"""Runs only the K3 / K3b launches of bench.py's attention roofline leg (local_config shape: B32 H16 d100 S315, prompt
221, stablemax, bf16, q/k pre-rotated), for rocprofv3 --pmc / --kernel-trace passes:
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU ... -d out -- python3 benchmarks/attn_pmc.py
benchmarks/attn_pmc_summary.py turns the counter CSVs into profiles/*_attn_pmc.json."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops  # noqa: E402

B, H, d, N, T = 32, 16, 100, 190, 94
if len(sys.argv) > 1:
    B, H, d = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
E, S, P = H * d, 31 + N + T, 31 + N
rot = int(d * 0.5)
torch.manual_seed(0)
qkv = (torch.rand(B, S, 3, E, device=dev) * 2 - 1).to(torch.bfloat16)
inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, device=dev).float() / rot))
cos_t, sin_t = ops.rope_tables(inv, S)
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
ops.rope_inplace(qkv.view(B, S, 3 * E)[:, :, :2 * E], 2 * H, rot, cos_t, sin_t)
o, lse = ops.prefix_attn_fwd(q, k, v, H, 0, P, cos_t, sin_t, True, save_lse=True)
do = torch.randn_like(o)
for _ in range(6):
    ops.prefix_attn_fwd(q, k, v, H, 0, P, cos_t, sin_t, True, save_lse=True)
    ops.prefix_attn_bwd(q, k, v, o, do, lse, H, 0, P, cos_t, sin_t, True)
torch.cuda.synchronize()
