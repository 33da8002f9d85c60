#!/usr/bin/env python3
"""Where should the decoder layers' optimizer updates run?  MODE=plain: inside step(); MODE=defer: at the top of the next
forward, beside the encoder forward (the product default); MODE=tail: when the decoder's backward has been issued, on the side
stream beside the ENCODER BACKWARD (about 190 small latency-bound launches that leave the chip mostly idle).  One GPU."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import multimodal_amd as mm  # noqa: E402
from multimodal_amd import functional as FN  # noqa: E402
from multimodal_amd.dp import DataParallel  # noqa: E402
from multimodal_amd.optim import OrthoGradAdamW  # noqa: E402

MODE = os.environ.get("MODE", "defer")
dev = torch.device("cuda:0")
c = bench.CONFIGS["local_config"]
mm.set_compute_dtype(torch.bfloat16)
model = bench.build_model(c, dev).train()
dp = DataParallel(model)
opt = OrthoGradAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, ortho=True, ortho_eps=1e-8, rescale=True)
if MODE == "defer":
    opt.defer_into_next_forward(model, model.decoder.layers)
ir, nmr, tgt = bench.synth_batch(c, 32, dev, 1234)
crit = torch.nn.CrossEntropyLoss(ignore_index=0)
dec_params = [p for l in model.decoder.layers for p in l.parameters()]


def tail(_g):
    main, side = FN.enter_side(dev)
    opt.begin_step()
    with torch.cuda.stream(side):
        opt.step_subset(dec_params, "tail")


if MODE == "tail":
    def hook(mod, inp, out):
        t = out[0] if isinstance(out, (tuple, list)) else out
        if t.requires_grad:
            t.register_hook(tail)
    model.encoder.register_forward_hook(hook)


def step():
    dp.zero_grad()
    logits = model(nmr, ir, tgt[:, :-1], None)
    loss = crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1))
    dp.scale_loss(loss).backward()
    dp.finish()
    opt.step()
    return loss


for _ in range(8):
    step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        loss = step()
    if MODE == "defer":
        opt.flush_deferred(wait=True)
    torch.cuda.synchronize()
    print(f"MODE={MODE}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step, loss {float(loss):.4f}", flush=True)
