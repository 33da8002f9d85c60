"""Edge cases of the hot path on the GPU (product modules, HIP through the C ABI) against tests/golden/edge.npz, which the
reference itself produced (oracle/gen_golden_edge.py): nmr_tokens=None, both truncation rules, B=1/T=1, a memory_proj
Linear (memory_dim != embed_dim), the length guard's ValueError, and [PAD]-ragged targets through CE(ignore_index=0).
Tolerances as in test_model_gpu.py: fp32 1e-4 on logits, 1e-3 on gradients; bf16 2e-2 on logits."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, elem_err, rel_err
from oracle import fill
from oracle.gen_golden_edge import EDGE, MEMPROJ, PAD_GRAD_KEYS, edge_inputs, memproj_inputs, padded_targets

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _reset_dtype():
    import multimodal_amd as mm
    yield
    mm.set_compute_dtype(torch.float32)


def _gold():
    return np.load(os.path.join(GOLDEN, "edge.npz"), allow_pickle=False)


def _model(c, dtype):
    import multimodal_amd as mm
    from multimodal_amd.models import MultiModalToSMILESModel
    m = MultiModalToSMILESModel(591, 1000, max_seq_length=c["max_seq"], max_nmr_length=c["max_nmr"],
                                max_memory_length=c["max_mem"], embed_dim=c["E"], num_heads=c["H"], num_layers=c["L"],
                                dropout=0.0, verbose=False, use_stablemax=c["sm"])
    fill.fill_state_dict_(m, seed=0)
    mm.set_compute_dtype(dtype)
    return m.to(DEV).eval()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["no_nmr", "truncate", "b1t1"])
def test_forward_edge_cases_vs_reference_golden(name, dtype):
    c = EDGE[name]
    m = _model(c, dtype)
    ir, nmr, tgt = edge_inputs(c)
    with torch.no_grad():
        lg = m(None if nmr is None else nmr.to(DEV), ir.to(DEV), tgt[:, :-1].to(DEV), None)
    assert lg.dtype == torch.float32
    assert rel_err(lg.cpu(), _gold()[name + ".logits"]) < (1e-4 if dtype == torch.float32 else 2e-2)
    if dtype == torch.float32:
        assert elem_err(lg.cpu(), _gold()[name + ".logits"]) < 1e-4


def test_padded_targets_loss_and_grads_vs_reference_golden():
    c = EDGE["padded"]
    m = _model(c, torch.float32)
    g = _gold()
    ir, nmr, _ = edge_inputs(c)
    tgt = padded_targets(c).to(DEV)
    lg = m(nmr.to(DEV), ir.to(DEV), tgt[:, :-1], None)
    loss = torch.nn.functional.cross_entropy(lg.reshape(-1, lg.size(-1)), tgt[:, 1:].reshape(-1), ignore_index=0)
    loss.backward()
    assert rel_err(lg.detach().cpu(), g["padded.logits"]) < 1e-4
    assert abs(loss.item() - float(g["padded.loss"])) < 1e-4 * abs(float(g["padded.loss"]))
    grads = dict(m.named_parameters())
    for k in PAD_GRAD_KEYS:
        assert rel_err(grads[k].grad.cpu(), g["padded.grad." + k]) < 1e-3, k
    assert rel_err(grads["decoder.smiles_embed.weight"].grad.cpu(), g["padded.grad.smiles_embed"]) < 1e-3


def test_memory_projection_and_length_guard_vs_reference_golden():
    from multimodal_amd.models import SMILESDecoder
    c = MEMPROJ
    d = SMILESDecoder(591, 1000, max_seq_length=c["max_seq"], max_nmr_length=c["max_nmr"], max_memory_length=c["max_mem"],
                      memory_dim=c["memory_dim"], embed_dim=c["E"], num_heads=c["H"], num_layers=c["L"], dropout=0.0,
                      verbose=False, use_stablemax=c["sm"])
    fill.fill_state_dict_(d, seed=0)
    d = d.to(DEV).eval()
    mem, nmr, tgt = memproj_inputs()
    g = _gold()
    with torch.no_grad():
        lg = d(tgt.to(DEV), mem.to(DEV), nmr.to(DEV))
    assert rel_err(lg.cpu(), g["memproj.logits"]) < 1e-4
    with pytest.raises(ValueError) as ei:
        d(torch.full((2, 95 + 256 - 31 - 4 + 1), 20, device=DEV), mem.to(DEV), nmr.to(DEV))
    assert str(ei.value) == str(g["guard.message"])


def test_layer_accepts_the_reference_mask_argument():
    """DecoderPromptLayer.forward(x, mask) with the reference's boolean [B,H,S,S] prefix-LM mask
    (transformer_decoder.py:295-298) equals the prompt_len form."""
    c = EDGE["b1t1"]
    m = _model(dict(c, B=2), torch.float32)
    lay = m.decoder.layers[0]
    B, Mp, T, E, H = 2, 9, 5, c["E"], c["H"]
    x = torch.from_numpy(fill.hash_uniform(B * (Mp + T) * E, 77).astype(np.float32)).reshape(B, Mp + T, E).to(DEV)
    mask = torch.ones(B, Mp + T, Mp + T, device=DEV).tril().bool()
    mask[:, :Mp, :Mp] = True
    mask = mask[:, None].repeat(1, H, 1, 1)
    with torch.no_grad():
        a = lay(x, mask)
        b = lay(x, None, prompt_len=Mp)
    assert torch.equal(a, b)
