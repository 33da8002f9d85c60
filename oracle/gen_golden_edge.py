#!/usr/bin/env python3
"""Edge-case golden vectors (oracle package: test infrastructure only).  Build container only: imports the
reference's ``models`` package, formula weights (oracle/fill.py), stores OUTPUTS in tests/golden/edge.npz:

  no_nmr      nmr_tokens=None (prompt = IR memory only; transformer_decoder.py:283-284)
  truncate    memory cut to max_memory_length=16 and NMR tokens cut to max_nmr_length=8 (:235-243)
  b1t1        batch 1, one target token
  memproj     SMILESDecoder alone with memory_dim != embed_dim (memory_proj is a Linear, :199,:259)
  padded      targets that contain [PAD]: CE with ignore_index=0 (train_autoregressive.py:672-674), loss + two grads

    python oracle/gen_golden_edge.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("MM_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
os.environ.setdefault("MPLBACKEND", "Agg")

from oracle import fill  # noqa: E402

BASE = dict(E=128, H=4, L=2, sm=True, max_seq=95, max_nmr=190, max_mem=32)
EDGE = {
    "no_nmr": dict(BASE, B=2, N=0, T=6),
    "truncate": dict(BASE, B=2, N=12, T=5, max_nmr=8, max_mem=16),
    "b1t1": dict(BASE, B=1, N=3, T=1),
    "padded": dict(BASE, B=3, N=6, T=9),
}
MEMPROJ = dict(E=128, H=4, L=2, sm=False, max_seq=95, max_nmr=190, max_mem=32, memory_dim=96, B=2, N=4, T=5)
PAD_GRAD_KEYS = ["decoder.out.bias", "decoder.layers.0.q.bias"]


def edge_inputs(c, seed=3):
    ir = fill.synth_ir(c["B"], 1000, seed)
    nmr = fill.synth_tokens(c["B"], c["N"], 0, 1000, seed) if c["N"] else None
    tgt = fill.synth_tokens(c["B"], c["T"] + 1, 15, 591, seed + 1)
    tgt[:, 0] = 12
    return ir, nmr, tgt


def padded_targets(c):
    _, _, tgt = edge_inputs(c)
    tgt[0, 5:] = 0                 # ragged rows: [PAD] tails of different lengths, one row unpadded
    tgt[1, 8:] = 0
    return tgt


def memproj_inputs(c=MEMPROJ):
    mem = torch.from_numpy(fill.hash_uniform(c["B"] * 31 * c["memory_dim"], 4242).astype(np.float32)).reshape(
        c["B"], 31, c["memory_dim"])
    nmr = fill.synth_tokens(c["B"], c["N"], 0, 1000, 9)
    tgt = fill.synth_tokens(c["B"], c["T"], 15, 591, 10)
    tgt[:, 0] = 12
    return mem, nmr, tgt


def main():
    sys.path.insert(0, REF)
    from models.multimodal_to_smiles import MultiModalToSMILESModel  # reference classes
    from models.transformer_decoder import SMILESDecoder
    store = {}
    for name, c in EDGE.items():
        m = MultiModalToSMILESModel(smiles_vocab_size=591, nmr_vocab_size=1000, max_seq_length=c["max_seq"],
                                    max_nmr_length=c["max_nmr"], max_memory_length=c["max_mem"], embed_dim=c["E"],
                                    num_heads=c["H"], num_layers=c["L"], dropout=0.0, verbose=False,
                                    use_stablemax=c["sm"]).eval()
        fill.fill_state_dict_(m, seed=0)
        ir, nmr, tgt = edge_inputs(c)
        if name == "padded":
            tgt = padded_targets(c)
            lg = m(nmr, ir, tgt[:, :-1], None)
            loss = torch.nn.functional.cross_entropy(lg.reshape(-1, lg.size(-1)), tgt[:, 1:].reshape(-1), ignore_index=0)
            loss.backward()
            store["padded.loss"] = np.array(loss.item(), dtype=np.float64)
            store["padded.logits"] = lg.detach().numpy()
            grads = dict(m.named_parameters())
            for k in PAD_GRAD_KEYS:
                store["padded.grad." + k] = grads[k].grad.numpy()
            store["padded.grad.smiles_embed"] = grads["decoder.smiles_embed.weight"].grad.numpy()
        else:
            with torch.no_grad():
                store[name + ".logits"] = m(nmr, ir, tgt[:, :-1], None).numpy()
    c = MEMPROJ
    d = SMILESDecoder(smiles_vocab_size=591, nmr_vocab_size=1000, max_seq_length=c["max_seq"], max_nmr_length=c["max_nmr"],
                      max_memory_length=c["max_mem"], memory_dim=c["memory_dim"], embed_dim=c["E"], num_heads=c["H"],
                      num_layers=c["L"], dropout=0.0, verbose=False, use_stablemax=c["sm"]).eval()
    fill.fill_state_dict_(d, seed=0)
    mem, nmr, tgt = memproj_inputs()
    with torch.no_grad():
        store["memproj.logits"] = d(tgt, mem, nmr).numpy()
    # the length guard (:249-253): message text of the ValueError
    try:
        d(torch.full((2, 95 + 256 - 31 - 4 + 1), 20), mem, nmr)
        store["guard.message"] = np.array("")
    except ValueError as e:
        store["guard.message"] = np.array(str(e))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "edge.npz"), **store)
    print("wrote", sorted(store))


if __name__ == "__main__":
    main()
