#!/usr/bin/env python3
"""Loss sequence of the config-size train step with the decoder updates inside step() (MODE=plain) or deferred into the next
forward (MODE=defer): the two must agree bit for bit, step by step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import multimodal_amd as mm  # noqa: E402
from multimodal_amd.dp import DataParallel  # noqa: E402
from multimodal_amd.optim import OrthoGradAdamW  # noqa: E402
MODE = os.environ.get("MODE", "defer")
dev = torch.device("cuda:0")
c = bench.CONFIGS["local_config"]
mm.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = bench.build_model(c, dev).train()
dp = DataParallel(model)
opt = OrthoGradAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, ortho=True, ortho_eps=1e-8, rescale=True)
if MODE == "defer":
    opt.defer_into_next_forward(model, model.decoder.layers)
if os.environ.get("SYNC") == "1":
    model.register_forward_pre_hook(lambda m, a: opt.flush_deferred(wait=True))     # the forward waits for every deferred update at its top
if os.environ.get("SYNC") == "2":
    model.decoder.register_forward_pre_hook(lambda m, a: opt.flush_deferred(wait=True))     # the DECODER forward waits for all of them
ir, nmr, tgt = bench.synth_batch(c, 32, dev, 1234)
crit = torch.nn.CrossEntropyLoss(ignore_index=0)
torch.manual_seed(1)
out = []
mems = []
sums = []
def _chk(m, a):
    sums.append([float(sum(p.detach().double().sum() for p in l.parameters())) for l in model.decoder.layers])
    if len(sums) in (2, 3):
        for li in (1, 5, 6):
            print('step', len(sums), 'layer', li, ' '.join(f'{n.split(".")[0][:6]}.{n.split(".")[-1][0]}={float(p.detach().double().sum()):.9f}' for n, p in model.decoder.layers[li].named_parameters()), flush=True)
model.decoder.register_forward_pre_hook(_chk)
model.encoder.register_forward_hook(lambda m, a, o: mems.append((o[0] if isinstance(o, (tuple, list)) else o).detach().double().sum()))
for i in range(int(os.environ.get("STEPS", "12"))):
    dp.zero_grad()
    logits = model(nmr, ir, tgt[:, :-1], None)
    loss = crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1))
    dp.scale_loss(loss).backward()
    dp.finish()
    opt.step()
    out.append(loss.detach())
print(MODE, " ".join(f"{float(x):.6f}" for x in out), flush=True)
print("decoder per-layer parameter sums at step 3:", " ".join(f"{x:.7f}" for x in sums[2]), flush=True)
print("decoder per-layer parameter sums at step 4:", " ".join(f"{x:.7f}" for x in sums[3]), flush=True)
print("encoder-output sums", " ".join(f"{float(x):.5f}" for x in mems[:6]), flush=True)
