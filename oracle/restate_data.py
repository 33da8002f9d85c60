"""CPU restatement of the reference's input pipeline (oracle package: TEST INFRASTRUCTURE ONLY -- imported by tests/,
never by the product).  PARITY UNPINNED: train_autoregressive.py needs wandb / rdkit / heavyball at import time, the
reference holds no fixture for this path, so these functions follow the source text line by line and nothing else pins
them.  Plain Python loops on purpose (small cases)."""
import json
from pathlib import Path

import numpy as np

from oracle.restate import encode_smiles


def dataset_items(data_dir, vocab, nmr_vocab, split="train", max_smiles_len=512, max_nmr_len=128):
    """train_autoregressive.py:256-282 (loading) and :325-361 (__getitem__), as lists instead of tensors."""
    d = Path(data_dir)
    sources = [line.strip() for line in open(d / f"src-{split}.txt")]                    # :257-258
    targets = [line.strip().replace(" ", "") for line in open(d / f"tgt-{split}.txt")]   # :259-261
    ir = None
    p = d / f"ir-{split}.npy"
    if p.exists():                                                                        # :266-282
        flat = np.fromfile(p, dtype="float32")
        ir = flat.reshape(len(sources), flat.shape[0] // len(sources))
    items = []
    for i in range(len(sources)):
        # :332-337 -> HF encode(add_special_tokens=True, max_length, truncation=True) over _tokenize's 510-token cut
        tgt = encode_smiles(targets[i], vocab, 512)
        if len(tgt) > max_smiles_len:
            tgt = tgt[: max_smiles_len - 1] + tgt[-1:]
        nmr = [nmr_vocab.get(tok, nmr_vocab["<UNK>"]) for tok in sources[i].split()]      # :343-345
        nmr = nmr[:max_nmr_len]                                                           # :346-347
        items.append((tgt, None if ir is None else ir[i].copy(), nmr))
    return items


def collate(items, nmr_vocab, pad_id=0):
    """train_autoregressive.py:372-411."""
    tw = max(len(t) for t, _, _ in items)                                                 # :382
    nw = max(len(n) for _, _, n in items)                                                 # :393
    tgt = np.array([t + [pad_id] * (tw - len(t)) for t, _, _ in items], dtype=np.int64).reshape(len(items), tw)
    nmr = np.array([n + [nmr_vocab["<PAD>"]] * (nw - len(n)) for _, _, n in items], dtype=np.int64).reshape(len(items), nw)
    irs = [x for _, x, _ in items if x is not None]                                       # :405-409
    return tgt, (np.stack(irs, 0) if irs else None), nmr


def vocab_sizes(tokenized_dir, n_smiles_vocab):
    """train_autoregressive.py:414-438."""
    with open(Path(tokenized_dir).parent / "vocab.json") as fh:
        v = json.load(fh)
    return n_smiles_vocab, max(v.values()) + 1, v
