"""Host time to ISSUE one train step (no sync) vs the GPU time of the step: is the step launch-bound?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import multimodal_amd as mm
from multimodal_amd.dp import DataParallel
from multimodal_amd.optim import OrthoGradAdamW

dev = torch.device("cuda", 0)
c = bench.CONFIGS["local_config"]
mm.set_compute_dtype(torch.bfloat16)
for B in (256, 32):
    model = bench.build_model(c, dev).train()
    dp = DataParallel(model)
    opt = OrthoGradAdamW(model.parameters(), lr=1e-4, eps=1e-8, weight_decay=0.01, ortho_eps=1e-8)
    ir, nmr, tgt = bench.synth_batch(c, B, dev, 1)
    crit = torch.nn.CrossEntropyLoss(ignore_index=0)

    def step():
        dp.zero_grad()
        logits = model(nmr, ir, tgt[:, :-1], None)
        crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1)).backward()
        dp.finish()
        opt.step()

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    hs, gs = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        hs.append((t1 - t0) * 1e3)
        gs.append((t2 - t0) * 1e3)
    print(f"B={B}: host issue {min(hs):.2f} ms (median {sorted(hs)[2]:.2f}), issue+drain {min(gs):.2f} ms", flush=True)
    del model, dp, opt
    torch.cuda.empty_cache()
