#!/usr/bin/env python3
"""Bitwise repeatability of the encoder's forward + backward (bf16, config batch): the parameter-gradient GEMMs (gemm_k, side
stream) run beside the main stream's K1b / LayerNorm / LayerScale kernels, whose arithmetic the compiler packs into v_pk_* f32
pairs.  Every run must reproduce the first run's bits."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd as mm  # noqa: E402
from multimodal_amd import functional as FN  # noqa: E402
from multimodal_amd.models import MultimodalSpectralEncoder  # noqa: E402
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
mm.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
enc = MultimodalSpectralEncoder(embed_dim=1600, verbose=False).to(dev).train()
ir = torch.rand(B, 1000, device=dev)
g = None
ref = None
bad = {}
for it in range(int(os.environ.get("RUNS", "12"))):
    for p in enc.parameters():
        p.grad = None
    torch.manual_seed(7)
    mem = enc(None, ir, None)
    if g is None:
        g = torch.randn_like(mem)
    mem.backward(g)
    FN.join_side_stream()
    torch.cuda.synchronize()
    cur = {"out": mem.detach().clone()}
    cur.update({n: p.grad.detach().clone() for n, p in enc.named_parameters() if p.grad is not None})
    if ref is None:
        ref = cur
        continue
    for n in ref:
        if not torch.equal(ref[n], cur[n]):
            bad[n] = bad.get(n, 0) + 1
print("runs differing from the first, per tensor:", dict(sorted(bad.items(), key=lambda kv: -kv[1])[:12]) if bad else "none", flush=True)
