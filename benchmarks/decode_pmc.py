#!/usr/bin/env python3
"""Runs only the K4 (decode_attn1_k) launches of bench.py's roofline_decode leg (same shape), for rocprofv3 --pmc passes:
    rocprofv3 --pmc FETCH_SIZE -d out_f -- python3 benchmarks/decode_pmc.py
    rocprofv3 --pmc WRITE_SIZE -d out_w -- python3 benchmarks/decode_pmc.py
benchmarks/pmc_summary.py turns the two counter CSVs into profiles/*_k4_pmc.json (per-launch HBM bytes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops  # noqa: E402

B, H, d, Smax = 1024, 8, 128, 238
E, rot, pos = H * d, 64, Smax - 1
dev = torch.device("cuda:0")
kc = (torch.rand(B, H, Smax, d, device=dev) - 0.5).to(torch.bfloat16)
vc = (torch.rand(B, H, Smax, d, device=dev) - 0.5).to(torch.bfloat16)
qkv = (torch.rand(B, 3 * E, device=dev) - 0.5).to(torch.bfloat16)
inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, device=dev).float() / rot))
cos_t, sin_t = ops.rope_tables(inv, Smax)
for _ in range(8):
    ops.decode_attn(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], kc, vc, H, rot, pos, cos_t, sin_t, False)
torch.cuda.synchronize()
