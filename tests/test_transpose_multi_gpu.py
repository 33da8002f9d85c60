"""mm_cast_transpose_multi: every transposed bf16 operand copy refreshed by one launch after an optimizer step.
Bit-exact against torch's w.t().to(bf16) (reference: autograd of nn.Linear, models/transformer_decoder.py:159-163 -- the copy
only changes the operand layout of dX = dY . W, not its values)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_refresh_after_optimizer_step_equals_fresh_transposes():
    from multimodal_amd import functional as FN
    from multimodal_amd.optim import OrthoGradAdamW
    d = torch.device("cuda:0")
    torch.manual_seed(0)
    shapes = [(256, 128), (136, 72), (64, 200), (128, 128), (128, 128), (128, 128)]
    ps = [torch.nn.Parameter(torch.randn(*s, device=d)) for s in shapes]
    odd = torch.nn.Parameter(torch.randn(30, 20, device=d))             # not a multiple of 8: stays on the per-use path
    opt = OrthoGradAdamW(ps + [odd], lr=1e-2)
    dt = torch.bfloat16
    groups = [(ps[0],), (ps[1],), (ps[2],), (ps[3], ps[4], ps[5]), (odd,)]
    FN.cached_operand((ps[0],), dt)                                      # a bf16 operand copy exists for ps[0] only: bf16 source
    FN.cached_operand((ps[3], ps[4], ps[5]), dt)                         # and for the stacked group
    for g in groups:
        FN.cached_operand(g, dt, transpose=True)
    for step in range(2):
        for p in ps + [odd]:
            p.grad = torch.randn_like(p)
        opt.step()                                                       # -> refresh_transposed_operands()
        for g in groups:
            key = tuple(id(p) for p in g) + (dt, True)
            ent = FN._operand_cache[key]
            fresh = all(s == (p._version, p.data_ptr()) for s, p in zip(ent[0], g))
            assert fresh == (g[0] is not odd)                            # refreshed in the optimizer, except the odd one
            want = torch.cat([p.detach() for p in g], 0).to(dt).t().contiguous()
            assert torch.equal(FN.cached_operand(g, dt, transpose=True), want)


def test_multi_launch_switch_off_is_identical(monkeypatch):
    from multimodal_amd import functional as FN
    d = torch.device("cuda:0")
    w = torch.nn.Parameter(torch.randn(64, 64, device=d))
    a = FN.cached_operand((w,), torch.bfloat16, transpose=True).clone()
    with torch.no_grad():
        w.mul_(1.5)
    monkeypatch.setattr(FN, "_TR_MULTI", False)
    assert FN.refresh_transposed_operands() == 0
    b = FN.cached_operand((w,), torch.bfloat16, transpose=True)
    assert torch.equal(b, (w.detach() .to(torch.bfloat16)).t()) and not torch.equal(a, b)
