"""The prefetching loader's device path: pinned staging buffers, copies on a copy stream, the consumer's stream waits on
an event -- the device batches equal the CPU collate (restatement), and feed one train step of the model."""
import numpy as np
import pytest
import torch

from oracle import restate, restate_data
from test_data_cpu import GOLDEN, SMILES, _tok, data_dir  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_device_batches_equal_collate_and_feed_a_train_step(data_dir):  # noqa: F811
    import os
    from multimodal_amd.data import PinnedBatchLoader, SpectralSmilesDataset, load_vocabularies
    from multimodal_amd.models import MultiModalToSMILESModel
    tok = _tok()
    ns, nn, nmr_vocab = load_vocabularies(data_dir, tok)
    ds = SpectralSmilesDataset(data_dir, tok, nmr_vocab, "train", 95, 190)
    vocab = restate.read_vocab(os.path.join(GOLDEN, "vocab.txt"))
    want = restate_data.dataset_items(data_dir, vocab, nmr_vocab, "train", 95, 190)
    ld = PinnedBatchLoader(ds, 4, device="cuda:0", depth=2)
    m = MultiModalToSMILESModel(ns, nn, 95, 190, 32, embed_dim=128, num_heads=4, num_layers=1, use_stablemax=True).to("cuda:0")
    for epoch in range(2):                                           # second epoch: every pinned buffer set is reused
        for step, (tb, ib, nb, _) in enumerate(ld):
            idx = list(range(step * 4, min(len(SMILES), step * 4 + 4)))
            wt, wi, wn = restate_data.collate([want[i] for i in idx], nmr_vocab)
            assert tb.is_cuda and np.array_equal(tb.cpu().numpy(), wt) and np.array_equal(nb.cpu().numpy(), wn)
            assert np.array_equal(ib.cpu().numpy(), wi)
            if tb.shape[1] > 2:
                logits = m(nb, ib, tb[:, :-1], None)
                loss = torch.nn.functional.cross_entropy(logits.reshape(-1, ns), tb[:, 1:].reshape(-1), ignore_index=0)
                loss.backward()
                assert torch.isfinite(loss)
