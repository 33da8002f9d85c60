"""GPU parity of K3 (prefix-LM attention fwd/bwd) and K4 (KV-cache decode step) against the CPU oracle
(oracle/restate.py: apply_partial_rope + attention_core + prefix_lm_mask)."""
import pytest
import torch

from conftest import rel_err
from oracle import restate

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def _oracle(q, k, v, H, rot, P, stablemax):
    B, S, E = q.shape
    d = E // H
    qh, kh, vh = (t.view(B, S, H, d).transpose(1, 2) for t in (q, k, v))
    qh, kh = restate.apply_partial_rope(qh, kh, rot)
    mask = restate.prefix_lm_mask(B, P, S - P)[:, None]
    a = restate.attention_core(qh, kh, vh, mask, stablemax)
    return a.transpose(1, 2).contiguous().view(B, S, E)


CASES = [  # B, H, d, S, P
    (2, 4, 32, 33, 21), (2, 2, 100, 47, 38), (1, 2, 128, 150, 100), (2, 2, 192, 70, 36), (1, 2, 512, 40, 36),
    (1, 16, 100, 315, 221), (2, 3, 64, 17, 0), (1, 1, 32, 5, 5), (3, 2, 128, 65, 64),
    # head_dim 512 (attention_wide.hip on the pre-rotated bf16 path): the real_config_mup sequence, a causal-only head,
    # lengths that are / are not multiples of the 32-row tiles and of the 64-row workgroup blocks
    (1, 2, 512, 315, 221), (2, 1, 512, 130, 0), (1, 1, 512, 64, 64), (1, 3, 512, 97, 33),
    # >= 128 heads and a head that fits LDS: the whole-head-resident kernels (attention_res.hip) on the pre-rotated bf16 path --
    # the local/real_config head (d 100: three 32-column steps + one of 16, 7 output tiles), the nocat head (d 128, softmax in
    # its config), narrower geometries, causal-only and prompt-only masks, lengths off the 16- and 32-row grids
    (8, 16, 100, 315, 221), (16, 8, 128, 238, 159), (4, 32, 32, 70, 40), (2, 64, 64, 97, 33), (4, 32, 96, 80, 0),
    (2, 64, 112, 130, 130), (8, 16, 100, 64, 3),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("stablemax", [True, False])
@pytest.mark.parametrize("case", CASES)
def test_prefix_attn_fwd_bwd(case, stablemax, dtype):
    from multimodal_amd import ops
    B, H, d, S, P = case
    E = H * d
    rot = int(d * 0.5)
    rot -= rot % 2
    dev = torch.device("cuda:0")
    qkv = _rand(B, S, 3, E, seed=1, scale=1.5).to(dtype)
    q, k, v = (qkv[:, :, i].float().requires_grad_(True) for i in range(3))
    ref = _oracle(q, k, v, H, rot, P, stablemax)
    d_o = _rand(B, S, E, seed=2).to(dtype)
    ref.backward(d_o.float())
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))
    cos_t, sin_t = ops.rope_tables(inv.to(dev), S)
    qkv_d = qkv.to(dev)
    qd, kd, vd = qkv_d[:, :, 0], qkv_d[:, :, 1], qkv_d[:, :, 2]        # strided views of the fused projection
    o, lse = ops.prefix_attn_fwd(qd, kd, vd, H, rot, P, cos_t, sin_t, stablemax, save_lse=True)
    torch.cuda.synchronize()
    assert rel_err(o.float().cpu(), ref.detach()) < TOL[dtype]
    dqkv = ops.prefix_attn_bwd(qd, kd, vd, o, d_o.to(dev), lse, H, rot, P, cos_t, sin_t, stablemax)
    torch.cuda.synchronize()
    gtol = 2e-4 if dtype == torch.float32 else 3e-2
    assert rel_err(dqkv[:, :, 2].float().cpu(), v.grad) < gtol
    assert rel_err(dqkv[:, :, 0].float().cpu(), q.grad) < gtol
    assert rel_err(dqkv[:, :, 1].float().cpu(), k.grad) < gtol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("stablemax", [True, False])
@pytest.mark.parametrize("case", [(3, 4, 32, 20, 12), (2, 2, 100, 40, 33), (2, 8, 128, 60, 50), (1, 2, 192, 30, 20), (1, 2, 512, 24, 18)])
def test_decode_attn_matches_full_recompute(case, stablemax, dtype):
    """K4: cache fill over the prompt + one decode step per target position == row `pos` of the oracle's full
    attention (SURVEY.md A9 legality argument)."""
    from multimodal_amd import ops
    B, H, d, S, P = case
    E = H * d
    rot = int(d * 0.5)
    rot -= rot % 2
    dev = torch.device("cuda:0")
    qkv = _rand(B, S, 3, E, seed=3, scale=1.5).to(dtype)
    q, k, v = (qkv[:, :, i].float() for i in range(3))
    ref = _oracle(q, k, v, H, rot, P, stablemax)
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))
    Smax = S + 3
    cos_t, sin_t = ops.rope_tables(inv.to(dev), Smax)
    qkv_d = qkv.to(dev)
    kc = torch.zeros(B, H, Smax, d, dtype=dtype, device=dev)
    vc = torch.zeros_like(kc)
    pre = qkv_d[:, :P].contiguous()
    ops.kv_cache_fill(pre[:, :, 1], pre[:, :, 2], kc, vc, H, rot, cos_t, sin_t)
    for pos in range(P, S):
        o = ops.decode_attn(qkv_d[:, pos, 0], qkv_d[:, pos, 1], qkv_d[:, pos, 2], kc, vc, H, rot, pos, cos_t, sin_t, stablemax)
        assert rel_err(o.float().cpu(), ref[:, pos]) < TOL[dtype], pos


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("stablemax", [True, False])
@pytest.mark.parametrize("case", [(2, 3, 128, 9), (1, 2, 100, 7), (1, 1, 512, 6)])
def test_decode_attn_from_an_empty_poisoned_cache(case, stablemax, dtype):
    """K4 from position 0 on (no cached row at all, then 1, 2, 3 ... rows: fewer than one 4-row iteration, ragged last
    iterations) with every cache element NaN before it is written: rows at or beyond `pos` must never reach the result
    (the one-pass kernel clamps the rows of its last iterations and weighs them 0)."""
    from multimodal_amd import ops
    B, H, d, S = case
    E = H * d
    rot = int(d * 0.5)
    rot -= rot % 2
    dev = torch.device("cuda:0")
    qkv = _rand(B, S, 3, E, seed=5, scale=1.5).to(dtype)
    q, k, v = (qkv[:, :, i].float() for i in range(3))
    ref = _oracle(q, k, v, H, rot, 0, stablemax)                       # no prefix: causal
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))
    Smax = S + 2
    cos_t, sin_t = ops.rope_tables(inv.to(dev), Smax)
    qkv_d = qkv.to(dev)
    kc = torch.full((B, H, Smax, d), float("nan"), dtype=dtype, device=dev)
    vc = torch.full_like(kc, float("nan"))
    for pos in range(S):
        o = ops.decode_attn(qkv_d[:, pos, 0], qkv_d[:, pos, 1], qkv_d[:, pos, 2], kc, vc, H, rot, pos, cos_t, sin_t, stablemax)
        assert torch.isfinite(o).all(), pos
        assert rel_err(o.float().cpu(), ref[:, pos]) < TOL[dtype], pos
    assert torch.isnan(kc[:, :, S:]).all() and torch.isfinite(kc[:, :, :S]).all()


@pytest.mark.parametrize("stablemax", [True, False])
@pytest.mark.parametrize("case", CASES)
def test_prefix_attn_prerotated_bf16_fast_path(case, stablemax):
    """The product path: q,k rotated once by mm_rope_inplace, attention called with rot=0 (bf16 fast kernels),
    gradients un-rotated by the inverse pre-pass."""
    from multimodal_amd import ops
    B, H, d, S, P = case
    E = H * d
    rot = int(d * 0.5)
    rot -= rot % 2
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16
    qkv = _rand(B, S, 3, E, seed=11, scale=1.5).to(dtype)
    q, k, v = (qkv[:, :, i].float().requires_grad_(True) for i in range(3))
    ref = _oracle(q, k, v, H, rot, P, stablemax)
    d_o = _rand(B, S, E, seed=12).to(dtype)
    ref.backward(d_o.float())
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))
    cos_t, sin_t = ops.rope_tables(inv.to(dev), S)
    qkv_d = qkv.to(dev)
    qd, kd, vd = qkv_d[:, :, 0], qkv_d[:, :, 1], qkv_d[:, :, 2]
    ops.rope_inplace(qd, H, rot, cos_t, sin_t)
    ops.rope_inplace(kd, H, rot, cos_t, sin_t)
    o, lse = ops.prefix_attn_fwd(qd, kd, vd, H, 0, P, cos_t, sin_t, stablemax, save_lse=True)
    assert rel_err(o.float().cpu(), ref.detach()) < 2e-2
    dqkv = ops.prefix_attn_bwd(qd, kd, vd, o, d_o.to(dev), lse, H, 0, P, cos_t, sin_t, stablemax)
    ops.rope_inplace(dqkv[:, :, 0], H, rot, cos_t, sin_t, inverse=True)
    ops.rope_inplace(dqkv[:, :, 1], H, rot, cos_t, sin_t, inverse=True)
    torch.cuda.synchronize()
    assert rel_err(dqkv[:, :, 2].float().cpu(), v.grad) < 3e-2
    assert rel_err(dqkv[:, :, 0].float().cpu(), q.grad) < 3e-2
    assert rel_err(dqkv[:, :, 1].float().cpu(), k.grad) < 3e-2
