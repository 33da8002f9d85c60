"""mm_lt_wgrad (the weight-gradient product on the hipBLASLt library, algorithm picked by a sweep) against a float32 product of
the same bf16 operands: every algorithm the sweep may pick accumulates in float32, so the bound is the summation-order bound of
an fp32 reduction over `rows` terms (2e-2 of the row scale would be the bf16 bar; this result is f32 and is held to 1e-3)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(dy, x):
    return dy.float().t() @ x.float()


@pytest.mark.parametrize("rows,n,k", [(2048, 1024, 1024), (10080, 1600, 1600), (4099, 1280, 1032), (3008, 6400, 1600)])
@pytest.mark.parametrize("tune", [False, True])
def test_lt_wgrad_matches_fp32_product(rows, n, k, tune):
    from multimodal_amd import ops
    g = torch.Generator(device="cuda").manual_seed(rows + n + k)
    dy = torch.randn(rows, n, device="cuda", generator=g).to(torch.bfloat16)
    x = torch.randn(rows, k, device="cuda", generator=g).to(torch.bfloat16)
    got = ops.lt_wgrad(dy, x, tune=tune)
    again = ops.lt_wgrad(dy, x, tune=tune)          # the remembered pick: same bits as the call that made it
    ref = _ref(dy, x)
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 1e-3 * scale
    assert torch.equal(got, again)


def test_lt_wgrad_strided_operands_and_output_slot():
    """q/k/v share one dy buffer (a column slice: row stride 3n) and write adjacent rows of one gradient slot."""
    from multimodal_amd import ops
    rows, n, k = 4096, 1024, 1024
    g = torch.Generator(device="cuda").manual_seed(7)
    dy3 = torch.randn(rows, 3 * n, device="cuda", generator=g).to(torch.bfloat16)
    x = torch.randn(rows, k + 8, device="cuda", generator=g).to(torch.bfloat16)[:, :k]
    slot = torch.full((3 * n + 1, k), float("nan"), device="cuda")
    for i in range(3):
        ops.lt_wgrad(dy3[:, i * n:(i + 1) * n], x, out=slot[i * n:(i + 1) * n])
    ref = _ref(dy3, x)
    assert torch.isnan(slot[3 * n:]).all()           # nothing written past the slot
    assert (slot[:3 * n] - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()


def test_lt_wgrad_rejects_bad_operands():
    from multimodal_amd import ops
    a = torch.zeros(64, 32, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(TypeError):
        ops.lt_wgrad(a.float(), a)
    with pytest.raises(ValueError):
        ops.lt_wgrad(a, a[:32])
    with pytest.raises(ValueError):
        ops.lt_wgrad(a.t(), a.t())
