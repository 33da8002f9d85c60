#!/usr/bin/env python3
"""Encoder + prefill of generate() alone (max_len=2) at the nocat shape, B = 1024: for rocprofv3 --kernel-trace --stats."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd as mm  # noqa: E402
from bench import CONFIGS, build_model, synth_batch  # noqa: E402
dev = torch.device("cuda:0")
c = CONFIGS["local_config_nocat"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
m = build_model(c, dev).eval()
ir, nmr, _ = synth_batch(c, B, dev, 4321)
with mm.compute_dtype(torch.bfloat16):
    for i in range(N):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.generate(nmr, ir, max_len=2, early_stop=False)
        torch.cuda.synchronize(); print(f"call {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
