#!/usr/bin/env python3
"""Greedy decode at the nocat shape split into its parts: encoder + prefill (max_len=2) against the full 80 steps; per-step
wall time, and the host time to ISSUE a step (no synchronisation) against the GPU time of the same steps."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd as mm  # noqa: E402
from bench import CONFIGS, build_model, synth_batch  # noqa: E402

dev = torch.device("cuda:0")
c = CONFIGS["local_config_nocat"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = build_model(c, dev).eval()
ir, nmr, _ = synth_batch(c, B, dev, 4321)


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.generate(nmr, ir, max_len=n, early_stop=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t0) * 1e3


with mm.compute_dtype(torch.bfloat16):
    for n in (2, 80, 2, 80, 41):
        host, wall = run(n)
        print(f"max_len={n:3d}: host issue {host:7.2f} ms, wall {wall:7.2f} ms", flush=True)
    h2, w2 = run(2)
    h80, w80 = run(80)
    print(f"per step: host {(h80 - h2) / 78 * 1e3:.0f} us, wall {(w80 - w2) / 78 * 1e3:.0f} us; prefill+encoder {w2:.1f} ms", flush=True)
