#!/usr/bin/env python3
"""Is the GPU ever waiting for the host inside the train step?  UNPROFILED measurement: at every phase boundary of the step the
host clock (when the boundary is ISSUED) and a HIP event on the main stream (when the GPU REACHES it) are taken against a
common origin.  lag = gpu_time - host_time: while the lag is large the GPU has queued work and cannot be idle; where the lag
is ~0 the GPU caught up with the host and anything the host does next is GPU idle time.  Also reported: the step's wall time
and the time from the last backward kernel's completion to the first optimizer kernel (the holes rocprofv3 showed in round 2
under the profiler: 1.31 + 0.77 ms)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import multimodal_amd as mm  # noqa: E402
from multimodal_amd.dp import DataParallel  # noqa: E402
from multimodal_amd.optim import OrthoGradAdamW  # noqa: E402

dev = torch.device("cuda:0")
c = bench.CONFIGS["local_config"]
mm.set_compute_dtype(torch.bfloat16)
model = bench.build_model(c, dev).train()
dp = DataParallel(model)
opt = OrthoGradAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, ortho=True, ortho_eps=1e-8, rescale=True)
if os.environ.get("MM_OPT_DEFER", "1") == "1":
    opt.defer_into_next_forward(model, model.decoder.layers)
ir, nmr, tgt = bench.synth_batch(c, 32, dev, 1234)
crit = torch.nn.CrossEntropyLoss(ignore_index=0)
NAMES = ["step start", "encoder forward issued", "forward issued", "loss issued", "encoder backward begins", "backward returned", "finish() returned", "opt.step() returned"]
_rec = [None]


def _mark(i):
    rec = _rec[0]
    if rec is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        rec.append((i, time.perf_counter(), ev))


def _enc_hook(mod, inp, out):
    _mark(1)
    t = out[0] if isinstance(out, (tuple, list)) else out
    if t.requires_grad:
        t.register_hook(lambda g: _mark(4))


model.encoder.register_forward_hook(_enc_hook)


def step(rec=None):
    _rec[0] = rec
    mark = _mark
    mark(0)
    dp.zero_grad()
    logits = model(nmr, ir, tgt[:, :-1], None)
    mark(2)
    loss = crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1))
    mark(3)
    dp.scale_loss(loss).backward()
    mark(5)
    dp.finish()
    mark(6)
    opt.step()
    mark(7)
    _rec[0] = None


for _ in range(8):
    step()
torch.cuda.synchronize()
rows = []
for rep in range(6):
    torch.cuda.synchronize()
    origin = torch.cuda.Event(enable_timing=True)
    origin.record()
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    rec = []
    if os.environ.get("STEADY") == "1":      # marks of the THIRD step of a run of four: the GPU is behind the host from its start on
        step(); step()
    step(rec)
    step()                      # the next step keeps the queue full, as in the training loop
    torch.cuda.synchronize()
    rows.append({i: ((h - h0) * 1e3, origin.elapsed_time(ev)) for i, h, ev in rec})
# steady-state step time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
opt.flush_deferred(wait=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 20 * 1e3
med = lambda xs: sorted(xs)[len(xs) // 2]
out = {"ms_per_step": round(ms, 3), "boundaries": []}
for k, name in enumerate(NAMES):
    host = med([r[k][0] for r in rows])
    gpu = med([r[k][1] for r in rows])
    out["boundaries"].append({"boundary": name, "host_issue_ms": round(host, 3), "gpu_reached_ms": round(gpu, 3), "gpu_lag_ms": round(gpu - host, 3)})
print(json.dumps(out, indent=1))
