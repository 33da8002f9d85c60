#!/usr/bin/env python3
"""Host time to ISSUE the phases of one train step (no synchronisation inside the step) against the step's wall time: a step whose
issue time equals its wall time is host-bound, and faster kernels cannot show."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd as mm  # noqa: E402
from multimodal_amd.dp import DataParallel  # noqa: E402
from multimodal_amd.optim import OrthoGradAdamW  # noqa: E402
from bench import CONFIGS, build_model, synth_batch  # noqa: E402

dev = torch.device("cuda:0")
c = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "local_config"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
mm.set_compute_dtype(torch.bfloat16)
model = build_model(c, dev).train()
dp = DataParallel(model)
opt = OrthoGradAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, ortho=True, ortho_eps=1e-8, rescale=True)
ir, nmr, tgt = synth_batch(c, B, dev, 1234)
crit = torch.nn.CrossEntropyLoss(ignore_index=0)
acc = {}


def step(timed):
    t = [time.perf_counter()]
    dp.zero_grad(); t.append(time.perf_counter())
    logits = model(nmr, ir, tgt[:, :-1], None); t.append(time.perf_counter())
    loss = crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1)); t.append(time.perf_counter())
    dp.scale_loss(loss).backward(); t.append(time.perf_counter())
    dp.finish(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if timed:
        for nm, a, b in zip(("zero_grad", "forward", "loss", "backward", "finish", "opt.step"), t[:-1], t[1:]):
            acc[nm] = acc.get(nm, 0.0) + (b - a)


for _ in range(5):
    step(False)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    step(True)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_wall = time.perf_counter() - t0
print(f"per step: host issue {t_issue / n * 1e3:.2f} ms, wall {t_wall / n * 1e3:.2f} ms", flush=True)
print("host per phase (ms): " + ", ".join(f"{k} {v / n * 1e3:.2f}" for k, v in acc.items()), flush=True)
# the same with the GPU drained before every step: pure issue cost, no back-pressure from a full queue
acc.clear()
tot = 0.0
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(True)
    tot += time.perf_counter() - t0
print(f"drained before each step: host issue {tot / n * 1e3:.2f} ms", flush=True)
print("host per phase (ms): " + ", ".join(f"{k} {v / n * 1e3:.2f}" for k, v in acc.items()), flush=True)

# ---- where does the backward end on each stream?  (events: main stream just before it joins the side stream; side stream's end)
from multimodal_amd import functional as FN  # noqa: E402
ev = {}
orig_join = FN.join_side_stream


def join_marked():
    if "main" not in ev:
        ev["main"] = torch.cuda.Event(enable_timing=True); ev["main"].record(torch.cuda.current_stream())
        ev["side"] = torch.cuda.Event(enable_timing=True); ev["side"].record(FN.side_stream(dev))
    orig_join()


FN.join_side_stream = join_marked
res = []
for _ in range(6):
    ev.clear()
    torch.cuda.synchronize()
    s0 = torch.cuda.Event(enable_timing=True); s0.record()
    dp.zero_grad()
    logits = model(nmr, ir, tgt[:, :-1], None)
    loss = crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1))
    f_end = torch.cuda.Event(enable_timing=True); f_end.record()
    dp.scale_loss(loss).backward()
    dp.finish()
    b_end = torch.cuda.Event(enable_timing=True); b_end.record()
    opt.step()
    o_end = torch.cuda.Event(enable_timing=True); o_end.record()
    torch.cuda.synchronize()
    res.append((s0.elapsed_time(f_end), s0.elapsed_time(ev["main"]), s0.elapsed_time(ev["side"]), s0.elapsed_time(b_end), s0.elapsed_time(o_end)))
for r in res[2:]:
    print("ms from step start: forward end %.2f | main stream done with backward %.2f | side stream done %.2f | finish() done %.2f | optimizer done %.2f" % r, flush=True)
