#!/usr/bin/env python3
"""Golden-vector generator (oracle package: test infrastructure only).

Runs ONLY in the build container, where /root/reference exists: imports the
reference's own ``models`` package, fills it with the closed-form weights of
``oracle/fill.py`` and stores OUTPUTS (never weights, never source) under
``tests/golden/``.  The committed fixtures are what pins ``oracle/restate.py``.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz, *.json
"""
from __future__ import annotations

import json
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

# name -> model/config of the case.  (E,H) pairs cover every head_dim of configs/*.yaml:
# d=192 (test), d=100 (local/real), d=128 (nocat), d=512 (real_mup), plus a tiny one.
CASES = {
    "tiny_sm":     dict(E=128,  H=4, L=2, sm=True,  B=2, N=10, T=12, max_seq=95, max_nmr=190, max_mem=32),
    "d100_sm":     dict(E=200,  H=2, L=2, sm=True,  B=2, N=7,  T=9,  max_seq=95, max_nmr=190, max_mem=32),
    "d128_sdpa":   dict(E=256,  H=2, L=2, sm=False, B=2, N=10, T=12, max_seq=80, max_nmr=128, max_mem=64),
    "d192_sm":     dict(E=384,  H=2, L=1, sm=True,  B=2, N=5,  T=6,  max_seq=95, max_nmr=190, max_mem=32),
    "d512_sm":     dict(E=1024, H=2, L=1, sm=True,  B=2, N=5,  T=6,  max_seq=95, max_nmr=190, max_mem=32),
    # configs/test_config.yaml itself (BASELINE.json configs[0]): batch 8, forward + greedy on CPU
    "test_config": dict(E=768,  H=4, L=8, sm=True,  B=8, N=20, T=8,  max_seq=95, max_nmr=190, max_mem=32),
}
GRAD_KEYS = [
    "encoder.ir_backbone.downsample_layers.0.0.weight",
    "encoder.ir_backbone.downsample_layers.1.3.bias",
    "encoder.ir_backbone.stages.0.0.dwconv.weight",
    "encoder.ir_backbone.stages.0.0.dwconv.bias",
    "encoder.ir_backbone.stages.0.0.gamma",
    "encoder.ir_backbone.stages.0.0.norm.weight",
    "encoder.ir_backbone.stages.2.5.norm.bias",
    "encoder.ir_backbone.stages.3.2.pwconv1.bias",
    "encoder.ir_backbone.stages.3.2.dwconv.weight",
    "encoder.ir_backbone.norm.weight",
    "decoder.layers.0.q.bias",
    "decoder.layers.0.attn_norm.weight",
    "decoder.layers.1.ffn_w2.bias",
    "decoder.final_norm.weight",
    "decoder.out.bias",
]
SMILES = [
    "CCO", "c1ccccc1", "CC(=O)Oc1ccccc1C(=O)O", "C[C@H](N)C(=O)O", "[Na+].[Cl-]", "BrCCl",
    "C1CC1%12CC%12", "N#Cc1ccc(cc1)[N+](=O)[O-]", "F/C=C\\F", "O=C(O)C[C@@H](O)C", "CC>>CO", "c1cc[nH]c1*",
    "Zz?$~:", "",
]


def build_reference(c):
    sys.path.insert(0, REF)
    from models.multimodal_to_smiles import MultiModalToSMILESModel  # reference class
    m = MultiModalToSMILESModel(
        smiles_vocab_size=591, nmr_vocab_size=1000, max_seq_length=c["max_seq"], max_nmr_length=c["max_nmr"],
        max_memory_length=c["max_mem"], embed_dim=c["E"], num_heads=c["H"], num_layers=c["L"], dropout=0.0,
        verbose=False, use_stablemax=c["sm"])
    fill.fill_state_dict_(m, seed=0)
    return m


def inputs_for(c, seed=0):
    ir = fill.synth_ir(c["B"], 1000, seed)
    nmr = fill.synth_tokens(c["B"], c["N"], 0, 1000, seed)
    tgt = fill.synth_tokens(c["B"], c["T"] + 1, 15, 591, seed + 1)
    tgt[:, 0] = 12
    return ir, nmr, tgt


def reference_greedy(model, nmr, ir, max_len):
    """Drives the reference modules exactly as train_autoregressive.py:80-116 does (that file cannot be
    imported here: it needs wandb/rdkit at import time) and records the appended ids."""
    bsz = nmr.size(0)
    cur = torch.tensor([[12]] * bsz)
    seqs = [[12] for _ in range(bsz)]
    done = [False] * bsz
    with torch.no_grad():
        memory = model.encoder(None, ir, None)
        for _ in range(min(max_len, model.decoder.max_seq_length)):
            logits = model.decoder(tgt=cur, memory=memory, nmr_tokens=nmr)
            nxt = logits[:, -1:].argmax(dim=-1)
            for i in range(bsz):
                if not done[i]:
                    t = nxt[i].item()
                    seqs[i].append(t)
                    if t == 13:
                        done[i] = True
            if all(done):
                break
            cur = torch.cat([cur, nxt], dim=1)
    return seqs


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    meta = {"torch": torch.__version__, "cases": CASES}
    torch.set_num_threads(8)
    for name, c in CASES.items():
        print("case", name, flush=True)
        model = build_reference(c).eval()
        ir, nmr, tgt = inputs_for(c)
        store = {}
        taps = {}
        hooks = []
        bb = model.encoder.ir_backbone
        if name in ("tiny_sm", "d100_sm"):
            hooks.append(bb.downsample_layers[0].register_forward_hook(lambda m, i, o: taps.__setitem__("stem", o)))
            for s in range(4):
                if s:
                    hooks.append(bb.downsample_layers[s].register_forward_hook(
                        lambda m, i, o, s=s: taps.__setitem__(f"down{s}", o)))
                for j in (0, len(bb.stages[s]) - 1):
                    hooks.append(bb.stages[s][j].register_forward_hook(
                        lambda m, i, o, s=s, j=j: taps.__setitem__(f"stage{s}.{j}", o)))
            for i, lyr in enumerate(model.decoder.layers):
                hooks.append(lyr.register_forward_hook(lambda m, i_, o, i=i: taps.__setitem__(f"layer{i}", o)))
        with torch.no_grad():
            memory = model.encoder(None, ir, None)
            logits = model(nmr, ir, tgt[:, :-1], None)
        for h in hooks:
            h.remove()
        store["memory"] = memory[:2].numpy()
        store["logits"] = logits.numpy()
        for k, v in taps.items():
            store["tap." + k] = v.detach().numpy()

        if name in ("tiny_sm", "d128_sdpa", "d100_sm"):
            # backward fixture: eval-mode (DropPath off) CE loss, as train_autoregressive.py:968-977
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
                if k in GRAD_KEYS:
                    store["grad." + k] = g.numpy()
            with open(os.path.join(out_dir, f"{name}.gradsums.json"), "w") as fh:
                json.dump(sums, fh, indent=0, sort_keys=True)

        if name == "tiny_sm":
            # train-mode forward: DropPath draws torch.rand (convnext.py:15) in block order
            model.train()
            torch.manual_seed(2024)
            with torch.no_grad():
                store["logits_train_seed2024"] = model(nmr, ir, tgt[:, :-1], None).numpy()
            model.eval()

        if name in ("tiny_sm", "d128_sdpa", "test_config", "d100_sm"):
            gb = 8
            ir8 = fill.synth_ir(gb, 1000, 5)
            nmr8 = fill.synth_tokens(gb, c["N"], 0, 1000, 5)
            steps = 12 if name == "test_config" else 24
            seqs = reference_greedy(model, nmr8, ir8, steps)
            meta.setdefault("greedy", {})[name] = {"max_len": steps, "ids": seqs}
        np.savez_compressed(os.path.join(out_dir, f"{name}.npz"), **store)

    # tokenizer / indexing fixture: the pure pieces of models/smiles_tokenizer.py import fine
    sys.path.insert(0, REF)
    from models.smiles_tokenizer import BasicSmilesTokenizer, load_vocab
    vocab = load_vocab(os.path.join(REF, "vocab.txt"))
    bt = BasicSmilesTokenizer()
    tok = {}
    for s in SMILES:
        toks = bt.tokenize(s)[:510]
        tok[s] = [12] + [vocab.get(t, vocab.get("[UNK]")) for t in toks] + [13]
    meta["tokenizer"] = {"n_vocab": len(vocab), "special": {k: vocab[k] for k in ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")},
                         "encodings": tok}
    with open(os.path.join(out_dir, "meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print("done")


if __name__ == "__main__":
    main()
