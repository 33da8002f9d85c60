"""GPU parity at the shapes of configs/*.yaml themselves (tests/golden/cfg_*.npz, produced by the reference through
oracle/gen_golden_config.py): E1600 H16 L8 stablemax S=315 (local/real_config), E1024 H8 L4 softmax S=238 (nocat, plus
the full 80 greedy steps at B=8), E2048 H4 L8 d=512 (real_config_mup); backward fixtures for every head_dim; and the
benchmarked mode -- bf16 training -- checked at model level against the reference's fp32 gradients.
Tolerances (BASELINE.json north_star): fp32 1e-4 relative (max-norm AND elementwise |a-b| <= 1e-4 (|b| + rms b)) on
outputs; bf16 2e-2 on encoder output and logits; gradients 1e-3 fp32 / 6e-2 (L2) bf16; greedy ids equal on the fp32 path."""
import json
import os

import pytest
import torch

from conftest import GOLDEN, case_inputs, check_grads_against_golden, elem_err, l2_err, load_golden, rel_err, report
from oracle import fill
from test_model_gpu import DEV, build

pytestmark = pytest.mark.gpu


def _say(*a):
    """Achieved errors go into the run's terminal summary (conftest.report), so a regression inside a tolerance is visible."""
    report(*a)


@pytest.fixture(autouse=True)
def _reset_dtype():
    import multimodal_amd as mm
    yield
    mm.set_compute_dtype(torch.float32)
    mm.set_gemm_backend("auto")
    torch.cuda.empty_cache()


# every config-size case on the default back end; the headline shape also on "own" (every dense product of the decoder on
# libmmhip's large tiles with their fused epilogues -- the mode bench.py times as train_own_gemms_bf16), against the SAME golden
CASES_BACKENDS = [("cfg_local", "auto"), ("cfg_local", "own"), ("cfg_nocat", "auto"), ("cfg_mup", "auto"), ("cfg_mup", "own")]


def _backend(name):
    import multimodal_amd as mm
    mm.set_gemm_backend(name)


def _case(config_meta, golden_meta, name):
    return (config_meta if name.startswith("cfg_") else golden_meta)["cases"][name]


@pytest.mark.parametrize("name,backend", CASES_BACKENDS)
def test_config_size_forward_fp32(config_meta, name, backend):
    _backend(backend)
    case = config_meta["cases"][name]
    m = build(case)
    ir, nmr, tgt = case_inputs(case)
    g = load_golden(name)
    with torch.no_grad():
        mem = m.encoder(None, ir.to(DEV), None)
        logits = m(nmr.to(DEV), ir.to(DEV), tgt[:, :-1].to(DEV), None)
    mem, logits = mem[:1].float().cpu(), logits.cpu()
    _say(name, "fp32 memory", rel_err(mem, g["memory"]), elem_err(mem, g["memory"]), "logits", rel_err(logits, g["logits"]),
         elem_err(logits, g["logits"]))
    assert rel_err(mem, g["memory"]) < 1e-4 and elem_err(mem, g["memory"]) < 1e-4
    assert rel_err(logits, g["logits"]) < 1e-4 and elem_err(logits, g["logits"]) < 1e-4


@pytest.mark.parametrize("name,backend", CASES_BACKENDS)
def test_config_size_forward_bf16(config_meta, name, backend):
    _backend(backend)
    case = config_meta["cases"][name]
    m = build(case, torch.bfloat16)
    ir, nmr, tgt = case_inputs(case)
    g = load_golden(name)
    with torch.no_grad():
        mem = m.encoder(None, ir.to(DEV), None)
        logits = m(nmr.to(DEV), ir.to(DEV), tgt[:, :-1].to(DEV), None)
    mem, logits = mem[:1].float().cpu(), logits.float().cpu()
    _say(name, "bf16 memory", rel_err(mem, g["memory"]), "logits", rel_err(logits, g["logits"]))
    assert rel_err(mem, g["memory"]) < 2e-2                      # encoder output: north_star's bf16 bar
    assert rel_err(logits, g["logits"]) < BF16_LOGITS_TOL         # behind L decoder layers


BF16_LOGITS_TOL = 2e-2          # achieved at the config shapes: 1.1-1.3e-2


def _backward(m, case):
    ir, nmr, tgt = case_inputs(case)
    ir_d = ir.to(DEV).requires_grad_(True)
    tg = tgt.to(DEV)
    logits = m(nmr.to(DEV), ir_d, tg[:, :-1], None)
    loss = torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.size(-1)), tg[:, 1:].reshape(-1), ignore_index=0)
    loss.backward()
    torch.cuda.synchronize()
    return loss.item(), ir_d.grad.float().cpu(), {k: p.grad for k, p in m.named_parameters()}


@pytest.mark.parametrize("name,backend", CASES_BACKENDS + [("d192_sm", "auto"), ("d512_sm", "auto")])
def test_backward_fp32_every_head_dim_and_config_size(config_meta, golden_meta, name, backend):
    _backend(backend)
    case = _case(config_meta, golden_meta, name)
    m = build(case)
    g = load_golden(name if name.startswith("cfg_") else name + "_bwd")
    loss, gir, grads = _backward(m, case)
    _say("fp32 loss rel", abs(loss - float(g["loss"])) / abs(float(g["loss"])), "grad.ir", rel_err(gir, g["grad.ir"]),
         "worst grad max-norm", max(rel_err(grads[k[5:]].float().cpu()[torch.from_numpy(g[k + ".rows"])] if k + ".rows" in g
                                            else grads[k[5:]].float().cpu(), v)
                                    for k, v in g.items() if k.startswith("grad.") and k != "grad.ir" and not k.endswith(".rows")))
    assert abs(loss - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert rel_err(gir, g["grad.ir"]) < 1e-3
    assert check_grads_against_golden(grads, g, 1e-3, name) >= 10
    with open(os.path.join(GOLDEN, f"{name}.gradsums.json")) as fh:
        sums = json.load(fh)
    for k, (s, a) in sums.items():
        got = float(grads[k].double().abs().sum())
        assert abs(got - a) <= 2e-3 * a + 1e-9, (k, got, a)


@pytest.mark.parametrize("name,backend", [(n, "auto") for n in ("tiny_sm", "d100_sm", "d128_sdpa", "d192_sm", "d512_sm")] + CASES_BACKENDS)
def test_backward_bf16_model_level_vs_reference_golden(config_meta, golden_meta, name, backend):
    """The benchmarked mode: bf16 activations and GEMM operands, f32 masters / statistics / gradients.  Against the
    reference's fp32 loss and gradients: loss 1e-2; every gradient tensor 6e-2 in relative L2 and 8e-2 in max-norm; |grad|
    sums 3e-2.  Where the bar comes from: a gradient carries the bf16 rounding of the forward AND of the backward pass
    through up to 15 encoder blocks and 8 decoder layers.  Measured worst tensor (always a small one at the far end of
    the chain, e.g. the first block's 64x7 conv weight): L2 1.2-2.5e-2 on the 1-2 layer cases, 2.3-4.5e-2 at the config
    sizes, and at a config size it moves inside that range when nothing but the GEMM tile shapes / summation order change
    (three builds of this library gave 3.7, 3.3 and 4.5e-2 on cfg_local) -- rounding noise, not a systematic term."""
    _backend(backend)
    case = _case(config_meta, golden_meta, name)
    m = build(case, torch.bfloat16)
    small = not name.startswith("cfg_")
    g = load_golden(name + "_bwd" if name in ("d192_sm", "d512_sm") else name)
    loss, gir, grads = _backward(m, case)
    worst = max(rel_err(grads[k[5:]].float().cpu()[torch.from_numpy(g[k + ".rows"])] if k + ".rows" in g else grads[k[5:]].float().cpu(), v)
                for k, v in g.items() if k.startswith("grad.") and k != "grad.ir" and not k.endswith(".rows"))
    worst_l2 = max(l2_err(grads[k[5:]].float().cpu()[torch.from_numpy(g[k + ".rows"])] if k + ".rows" in g else grads[k[5:]].float().cpu(), v)
                   for k, v in g.items() if k.startswith("grad.") and k != "grad.ir" and not k.endswith(".rows"))
    _say(name, "bf16 loss rel", abs(loss - float(g["loss"])) / abs(float(g["loss"])), "grad.ir", rel_err(gir, g["grad.ir"]),
         "worst key max-norm", worst, "worst key L2", worst_l2)
    assert abs(loss - float(g["loss"])) < 1e-2 * abs(float(g["loss"]))
    assert rel_err(gir, g["grad.ir"]) < 8e-2 and l2_err(gir, g["grad.ir"]) < 6e-2
    assert check_grads_against_golden(grads, g, 8e-2, name, l2_tol=6e-2) >= 10
    with open(os.path.join(GOLDEN, f"{name}.gradsums.json")) as fh:
        sums = json.load(fh)
    bad = [(k, float(grads[k].double().abs().sum()), a) for k, (s, a) in sums.items()
           if abs(float(grads[k].double().abs().sum()) - a) > 3e-2 * a + 1e-9]
    assert not bad, bad[:5]


def test_greedy_nocat_80_steps_ids_equal_reference_fp32(config_meta):
    """configs/local_config_nocat.yaml shape, B=8, all 80 steps, KV cache vs the reference's recompute loop: equal ids."""
    from multimodal_amd.generate import ids_to_lists
    case = config_meta["cases"]["cfg_nocat"]
    gold = config_meta["greedy"]["cfg_nocat"]
    m = build(case)
    ir8 = fill.synth_ir(8, 1000, 5).to(DEV)
    nmr8 = fill.synth_tokens(8, case["N"], 0, 1000, 5).to(DEV)
    toks = m.generate(nmr8, ir8, max_len=gold["max_len"])
    got = ids_to_lists(toks.cpu())
    for r, (a, b) in enumerate(zip(got, gold["ids"])):
        first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), None)
        assert a == b, f"row {r}: first difference at step {first}"


def test_greedy_bf16_leaves_the_reference_only_at_near_ties(config_meta):
    """bf16 greedy (the mode behind the tokens/s figure) cannot be bit-identical to the fp32 reference; what it must
    satisfy: a row's ids equal the reference's up to a step where the reference's own decision was a near-tie
    (top-1 minus top-2 logit below NEAR_TIE on the fp32 path, teacher-forced on the reference ids)."""
    import multimodal_amd as mm
    NEAR_TIE = 0.15                 # bf16 logits carry ~2-3e-2 of the logit scale (~4): gaps well above that must hold
    case = config_meta["cases"]["cfg_nocat"]
    gold = torch.tensor(config_meta["greedy"]["cfg_nocat"]["ids"])           # [8, 81]
    ir8 = fill.synth_ir(8, 1000, 5).to(DEV)
    nmr8 = fill.synth_tokens(8, case["N"], 0, 1000, 5).to(DEV)
    m = build(case)
    with torch.no_grad():
        lg = m(nmr8, ir8, gold[:, :-1].to(DEV), None).float()                # [8, 80, 591], fp32 path
    top2 = lg.topk(2, dim=-1).values
    gap = (top2[..., 0] - top2[..., 1]).cpu()                                # gap[b, t]: decision that produced id t+1
    assert torch.equal(lg.argmax(-1).cpu(), gold[:, 1:])                     # teacher-forced fp32 == the reference ids
    mm.set_compute_dtype(torch.bfloat16)
    toks = m.generate(nmr8, ir8, max_len=80, early_stop=False).cpu()
    kept = []
    for r in range(8):
        diff = (toks[r, :81] != gold[r]).nonzero()
        first = int(diff[0]) if len(diff) else 81
        kept.append(first)
        if first < 81:
            assert gap[r, first - 1] < NEAR_TIE, (r, first, float(gap[r, first - 1]))
    _say("bf16 greedy: ids equal to the reference for", kept, "of 81 positions; min fp32 gap", float(gap.min()))
