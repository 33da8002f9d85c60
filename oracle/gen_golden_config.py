#!/usr/bin/env python3
"""Config-size golden vectors (oracle package: test infrastructure only; runs ONLY in the build container, where
/root/reference exists).  Same recipe as gen_golden.py -- the reference's own ``models`` package, formula weights
(oracle/fill.py), outputs only -- at the shapes of configs/*.yaml themselves:

    cfg_local   configs/local_config.yaml = real_config.yaml   E1600 H16 L8 stablemax  N=190 T=94  (S=315)
    cfg_nocat   configs/local_config_nocat.yaml                E1024 H8  L4 softmax    N=128 T=79  (S=238) + greedy, 80 steps, B=8
    cfg_mup     configs/real_config_mup.yaml                   E2048 H4  L8 stablemax  N=190 T=94  (d=512)
    d192_sm / d512_sm   backward fixtures of the small cases gen_golden.py stores forward-only

Stored per case (B=2): memory[:1], logits, loss, grad.ir, the GRAD_KEYS gradients, and |grad| sums of every parameter.

    python oracle/gen_golden_config.py        # writes tests/golden/cfg_*.npz, *_bwd.npz, *.gradsums.json, meta_config.json
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
os.environ.setdefault("MPLBACKEND", "Agg")

from oracle import fill  # noqa: E402
from oracle.gen_golden import CASES as SMALL, GRAD_KEYS, build_reference, inputs_for, reference_greedy  # noqa: E402

CONFIG_CASES = {
    "cfg_local": dict(E=1600, H=16, L=8, sm=True,  B=2, N=190, T=94, max_seq=95, max_nmr=190, max_mem=32),
    "cfg_nocat": dict(E=1024, H=8,  L=4, sm=False, B=2, N=128, T=79, max_seq=80, max_nmr=128, max_mem=64),
    "cfg_mup":   dict(E=2048, H=4,  L=8, sm=True,  B=2, N=190, T=94, max_seq=95, max_nmr=190, max_mem=32),
}
# layers 0 and 1 exist in every case; the deep cases also pin the last layer
EXTRA_GRAD_KEYS = ["decoder.layers.7.ffn_w1.bias", "decoder.layers.3.attn_norm.weight", "decoder.nmr_embed.weight"]


def backward_store(model, ir, nmr, tgt, store, sums_path):
    model.zero_grad()
    ir_g = ir.clone().requires_grad_(True)
    lg = model(nmr, ir_g, tgt[:, :-1], None)
    loss = torch.nn.functional.cross_entropy(lg.reshape(-1, lg.size(-1)), tgt[:, 1:].reshape(-1), ignore_index=0)
    loss.backward()
    store["loss"] = np.array(loss.item(), dtype=np.float64)
    store["grad.ir"] = ir_g.grad.numpy()
    sums = {}
    for k, p in model.named_parameters():
        g = p.grad
        sums[k] = [float(g.double().sum()), float(g.double().abs().sum())]
        if k in GRAD_KEYS or k in EXTRA_GRAD_KEYS:
            if k.endswith("nmr_embed.weight"):          # only the rows that were looked up carry gradient: keep 64 of them
                rows = torch.unique(nmr.reshape(-1))[:64]
                store["grad." + k + ".rows"] = rows.numpy()
                store["grad." + k] = g[rows].numpy()
            else:
                store["grad." + k] = g.numpy()
    with open(sums_path, "w") as fh:
        json.dump(sums, fh, indent=0, sort_keys=True)


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    torch.set_num_threads(8)
    meta = {"torch": torch.__version__, "cases": CONFIG_CASES, "greedy": {}}
    for name, c in CONFIG_CASES.items():
        print("case", name, flush=True)
        model = build_reference(c).eval()
        ir, nmr, tgt = inputs_for(c)
        store = {}
        with torch.no_grad():
            memory = model.encoder(None, ir, None)
            logits = model(nmr, ir, tgt[:, :-1], None)
        store["memory"] = memory[:1].numpy()
        store["logits"] = logits.numpy()
        backward_store(model, ir, nmr, tgt, store, os.path.join(out_dir, f"{name}.gradsums.json"))
        if name == "cfg_nocat":
            gb = 8
            ir8 = fill.synth_ir(gb, 1000, 5)
            nmr8 = fill.synth_tokens(gb, c["N"], 0, 1000, 5)
            seqs = reference_greedy(model, nmr8, ir8, 80)
            meta["greedy"][name] = {"max_len": 80, "ids": seqs}
        np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **store)
        del model
    for name in ("d192_sm", "d512_sm"):
        print("backward of", name, flush=True)
        c = SMALL[name]
        model = build_reference(c).eval()
        ir, nmr, tgt = inputs_for(c)
        store = {}
        backward_store(model, ir, nmr, tgt, store, os.path.join(out_dir, f"{name}.gradsums.json"))
        np.savez_compressed(os.path.join(out_dir, f"{name}_bwd.npz"), **store)
    with open(os.path.join(out_dir, "meta_config.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print("done")


if __name__ == "__main__":
    main()
