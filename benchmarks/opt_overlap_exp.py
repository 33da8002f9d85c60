#!/usr/bin/env python3
"""Experiment: the optimizer update of decoder layers 1.. deferred into the NEXT step's decoder forward (side stream, one event
per layer, a layer's forward waits for its own update) against the plain step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd as mm  # noqa: E402
from multimodal_amd import functional as FN  # noqa: E402
from multimodal_amd.dp import DataParallel  # noqa: E402
from multimodal_amd.optim import OrthoGradAdamW  # noqa: E402
from bench import CONFIGS, build_model, synth_batch  # noqa: E402

dev = torch.device("cuda:0")
c = CONFIGS["local_config"]
mm.set_compute_dtype(torch.bfloat16)
model = build_model(c, dev).train()
dp = DataParallel(model)
opt = OrthoGradAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, ortho=True, ortho_eps=1e-8, rescale=True)
ir, nmr, tgt = synth_batch(c, 32, dev, 1234)
crit = torch.nn.CrossEntropyLoss(ignore_index=0)
FIRST = int(sys.argv[1]) if len(sys.argv) > 1 else 1
layers = list(model.decoder.layers)
late = [list(l.parameters()) for l in layers[FIRST:]]
late_ids = set(id(p) for ps in late for p in ps)
early = [p for p in model.parameters() if id(p) not in late_ids]
events = {}
pending = [False]


def launch_deferred(mod, args):
    if not pending[0]:
        return
    pending[0] = False
    main, side = FN.enter_side(dev)
    with torch.cuda.stream(side):
        for li, ps in enumerate(late):
            for p in ps:
                p.grad = p._mm_view
            opt.step_subset(ps, ("late", li))
            for p in ps:
                p.grad = None
            ev = torch.cuda.Event(); ev.record(side)
            events[FIRST + li] = ev
    opt._begun = False
    opt._done = set()


def make_wait(idx):
    def wait(mod, args):
        ev = events.pop(idx, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
    return wait


def step_plain():
    dp.zero_grad()
    logits = model(nmr, ir, tgt[:, :-1], None)
    loss = crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1))
    dp.scale_loss(loss).backward()
    dp.finish()
    opt.step()


def step_overlap():
    dp.zero_grad()
    logits = model(nmr, ir, tgt[:, :-1], None)
    loss = crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1))
    dp.scale_loss(loss).backward()
    dp.finish()
    opt.begin_step()
    opt.step_subset(early, "early")
    pending[0] = True


def timeit(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"plain: {timeit(step_plain):.2f} ms/step", flush=True)
(model if os.environ.get('HOOK') == 'model' else model.decoder).register_forward_pre_hook(launch_deferred)   # HOOK=model: beside the encoder forward too
for i, l in enumerate(layers):
    if i >= FIRST:
        l.register_forward_pre_hook(make_wait(i))
print(f"overlap (layers {FIRST}.. deferred into the next forward): {timeit(step_overlap):.2f} ms/step", flush=True)
