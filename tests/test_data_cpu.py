"""Input pipeline (multimodal_amd.data) against the line-by-line restatement in oracle/restate_data.py:
integer / byte exact.  PARITY UNPINNED (documented in both files): the reference module cannot be imported here and
ships no fixtures for this path; the file formats are the reference's (raw float32 memmap without a header,
whitespace token lines, vocab.json)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import restate, restate_data

SMILES = ["CCO", "c1ccccc1", "CC(=O)Oc1ccccc1C(=O)O", "C [C@H] (N) C(=O)O", "[Na+].[Cl-]", "BrCCl", "Zz?$~:", "",
          "N#Cc1ccc(cc1)[N+](=O)[O-]", "F/C=C\\F", "C" * 40]
NMR_VOCAB = {"<PAD>": 0, "<UNK>": 1, "1HNMR": 2, "13CNMR": 3, "7.2": 4, "3.5": 5, "d": 6, "t": 7, "|": 8, "128.4": 9}


@pytest.fixture()
def data_dir(tmp_path):
    d = tmp_path / "tok" / "data"
    d.mkdir(parents=True)
    rng = np.random.default_rng(0)
    toks = [t for t in NMR_VOCAB if not t.startswith("<")] + ["99.9", "q"]          # two out-of-vocabulary tokens
    with open(d / "src-train.txt", "w") as fh:
        for i in range(len(SMILES)):
            n = [0, 1, 3, 17, 40][i % 5]
            fh.write(" ".join(rng.choice(toks, n)) + ("  \n" if i % 2 else "\n"))
    with open(d / "tgt-train.txt", "w") as fh:
        for s in SMILES:
            fh.write(s + "\n")
    mm = np.memmap(d / "ir-train.npy", dtype="float32", mode="w+", shape=(len(SMILES), 1000))   # the writer's form
    mm[:] = rng.random((len(SMILES), 1000), dtype=np.float32)
    mm.flush()
    del mm
    with open(tmp_path / "tok" / "vocab.json", "w") as fh:
        json.dump(NMR_VOCAB, fh)
    return d


def _tok():
    from multimodal_amd.models import SmilesTokenizer
    return SmilesTokenizer(os.path.join(GOLDEN, "vocab.txt"))


@pytest.mark.parametrize("max_smiles_len,max_nmr_len", [(512, 128), (12, 5)])
def test_dataset_and_collate_equal_the_restatement(data_dir, max_smiles_len, max_nmr_len):
    from multimodal_amd.data import SpectralSmilesDataset, collate_fn, load_vocabularies
    tok = _tok()
    ns, nn, nmr_vocab = load_vocabularies(data_dir, tok)
    assert (ns, nn, nmr_vocab) == restate_data.vocab_sizes(data_dir, 591)
    ds = SpectralSmilesDataset(data_dir, tok, nmr_vocab, "train", max_smiles_len, max_nmr_len)
    vocab = restate.read_vocab(os.path.join(GOLDEN, "vocab.txt"))
    want = restate_data.dataset_items(data_dir, vocab, nmr_vocab, "train", max_smiles_len, max_nmr_len)
    assert len(ds) == len(want) == len(SMILES)
    for i, (t, x, n) in enumerate(want):
        tt, (ir, none1), nm, none2 = ds[i]
        assert tt.dtype == torch.long and tt.tolist() == t and nm.tolist() == n and none1 is None and none2 is None
        assert ir.dtype == torch.float32 and np.array_equal(ir.numpy(), x)
        assert len(t) <= max_smiles_len and len(n) <= max_nmr_len and t[0] == 12 and t[-1] == 13
    for idx in ([0, 1, 2, 3], [7, 7], [10, 4, 9], list(range(len(SMILES)))):
        tb, ib, nb, none = collate_fn([ds[i] for i in idx], nmr_vocab)
        wt, wi, wn = restate_data.collate([want[i] for i in idx], nmr_vocab)
        assert none is None and np.array_equal(tb.numpy(), wt) and np.array_equal(nb.numpy(), wn) and np.array_equal(ib.numpy(), wi)


@pytest.mark.parametrize("world,shuffle", [(1, False), (1, True), (2, True), (4, False)])
def test_prefetching_loader_serves_the_collated_batches(data_dir, world, shuffle):
    from multimodal_amd.data import PinnedBatchLoader, SpectralSmilesDataset, load_vocabularies
    tok = _tok()
    _, _, nmr_vocab = load_vocabularies(data_dir, tok)
    ds = SpectralSmilesDataset(data_dir, tok, nmr_vocab, "train", 512, 128)
    vocab = restate.read_vocab(os.path.join(GOLDEN, "vocab.txt"))
    want = restate_data.dataset_items(data_dir, vocab, nmr_vocab, "train", 512, 128)
    B = 2
    loaders = [PinnedBatchLoader(ds, B, shuffle=shuffle, seed=5, rank=r, world=world, depth=2) for r in range(world)]
    for epoch in range(2):
        n = len(ds)
        order = torch.randperm(n, generator=torch.Generator().manual_seed(5 + epoch)).tolist() if shuffle else list(range(n))
        got = [list(ld) for ld in loaders]
        assert len({len(g) for g in got}) == 1                      # every rank takes the same number of steps
        seen = []
        for step in range(len(got[0])):
            chunk = order[step * B * world: (step + 1) * B * world]
            per = len(chunk) // world
            for r in range(world):
                idx = chunk[r * per: (r + 1) * per]
                tb, ib, nb, none = got[r][step]
                wt, wi, wn = restate_data.collate([want[i] for i in idx], nmr_vocab)
                assert tb.dtype == torch.int64 and np.array_equal(tb.numpy(), wt) and np.array_equal(nb.numpy(), wn)
                assert np.array_equal(ib.numpy(), wi)
                seen += idx
        assert len(seen) == len(set(seen)) and len(seen) >= n - (B * world - 1)


def test_missing_ir_file_and_drop_last(data_dir):
    from multimodal_amd.data import PinnedBatchLoader, SpectralSmilesDataset, load_vocabularies
    os.remove(data_dir / "ir-train.npy")
    tok = _tok()
    _, _, nmr_vocab = load_vocabularies(data_dir, tok)
    ds = SpectralSmilesDataset(data_dir, tok, nmr_vocab)
    assert ds[0][1] == (None, None)
    batches = list(PinnedBatchLoader(ds, 4, drop_last=True))
    assert len(batches) == len(SMILES) // 4 and all(b[1] is None and b[0].shape[0] == 4 for b in batches)


def test_unreadable_ir_file_is_a_warning_not_a_crash(data_dir, capsys):
    """train_autoregressive.py:285-287: a float count that does not divide into the samples (numpy's reshape raises) is
    caught, warned about, and training goes on with ir_data = None."""
    from multimodal_amd.data import SpectralSmilesDataset, load_vocabularies
    with open(data_dir / "ir-train.npy", "ab") as fh:
        fh.write(np.zeros(7, dtype=np.float32).tobytes())
    tok = _tok()
    _, _, nmr_vocab = load_vocabularies(data_dir, tok)
    ds = SpectralSmilesDataset(data_dir, tok, nmr_vocab)
    assert ds.ir_data is None and ds[0][1] == (None, None)
    assert "[Warning] Failed to load IR data" in capsys.readouterr().out
