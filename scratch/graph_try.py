"""Experiment: capture one whole train step (fwd + CE + bwd + optimizer) in a HIP graph and time its replay against the
eager step.  (lr / bias corrections are baked in here: timing experiment only.)"""
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
cfgname = sys.argv[1] if len(sys.argv) > 1 else "local_config"
c = bench.CONFIGS[cfgname]
mm.set_compute_dtype(torch.bfloat16)
model = bench.build_model(c, dev).train()
dp = DataParallel(model)
opt = OrthoGradAdamW(model.parameters(), lr=1e-4, eps=1e-8, weight_decay=0.01, ortho_eps=1e-8)
ir, nmr, tgt = bench.synth_batch(c, 32, dev, 1)
crit = torch.nn.CrossEntropyLoss(ignore_index=0)
loss_buf = torch.zeros((), device=dev)


def step():
    dp.zero_grad()
    logits = model(nmr, ir, tgt[:, :-1], None)
    loss = crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1))
    loss.backward()
    dp.finish()
    opt.step()
    loss_buf.copy_(loss.detach())


def timeit(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(5):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print(f"{cfgname}: eager {timeit(step):.3f} ms/step, loss {loss_buf.item():.4f}", flush=True)
g = torch.cuda.CUDAGraph()
print("capturing", flush=True)
with torch.cuda.graph(g):
    step()
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print(f"{cfgname}: graph replay {timeit(g.replay):.3f} ms/step, loss {loss_buf.item():.4f}", flush=True)
