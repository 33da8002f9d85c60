"""End-to-end GPU parity of the product modules (multimodal_amd.models, HIP path through the C ABI) against
(1) the committed golden vectors produced by the reference itself and (2) the CPU oracle on the same inputs.
Tolerances (BASELINE.json north_star): fp32 1e-4 relative on encoder/attention outputs, bf16 2e-2; greedy ids equal."""
import json
import os

import pytest
import torch

from conftest import GOLDEN, case_inputs, cfg_of, elem_err, load_golden, rel_err
from oracle import fill, restate
from oracle.shapes import formula_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(case, dtype=torch.float32):
    import multimodal_amd as mm
    from multimodal_amd.models import MultiModalToSMILESModel
    m = MultiModalToSMILESModel(591, 1000, max_seq_length=case["max_seq"], max_nmr_length=case["max_nmr"],
                                max_memory_length=case["max_mem"], embed_dim=case["E"], num_heads=case["H"],
                                num_layers=case["L"], dropout=0.0, verbose=False, use_stablemax=case["sm"])
    fill.fill_state_dict_(m, seed=0)
    mm.set_compute_dtype(dtype)
    return m.to(DEV).eval()


@pytest.fixture(autouse=True)
def _reset_dtype():
    import multimodal_amd as mm
    yield
    mm.set_compute_dtype(torch.float32)
    mm.set_gemm_backend("auto")


def test_state_dict_loads_strict(golden_meta):
    case = golden_meta["cases"]["tiny_sm"]
    m = build(case)
    sd = formula_state_dict(cfg_of(case))
    m.load_state_dict(sd, strict=True)            # reference-shaped checkpoint: same keys, same shapes


@pytest.mark.parametrize("name", ["tiny_sm", "d100_sm", "d128_sdpa", "d192_sm", "d512_sm", "test_config"])
def test_forward_fp32_vs_reference_golden(golden_meta, name):
    case = golden_meta["cases"][name]
    m = build(case)
    ir, nmr, tgt = case_inputs(case)
    g = load_golden(name)
    with torch.no_grad():
        mem = m.encoder(None, ir.to(DEV), None)
        logits = m(nmr.to(DEV), ir.to(DEV), tgt[:, :-1].to(DEV), None)
    assert rel_err(mem[:2].float().cpu(), g["memory"]) < 1e-4 and elem_err(mem[:2].float().cpu(), g["memory"]) < 1e-4
    assert rel_err(logits.cpu(), g["logits"]) < 1e-4 and elem_err(logits.cpu(), g["logits"]) < 1e-4


# north_star's bf16 bar is 2e-2 on encoder/attention outputs; the logits behind L decoder layers are held to it too
# (achieved 0.7-1.2e-2 here, 1.1-1.3e-2 at the config shapes in tests/test_config_gpu.py).
BF16_LOGITS_TOL = {"tiny_sm": 2e-2, "d100_sm": 2e-2, "d128_sdpa": 2e-2, "d192_sm": 2e-2, "d512_sm": 2e-2}


@pytest.mark.parametrize("name", ["tiny_sm", "d100_sm", "d128_sdpa", "d192_sm", "d512_sm"])
def test_forward_bf16_vs_reference_golden(golden_meta, name):
    case = golden_meta["cases"][name]
    m = build(case, torch.bfloat16)
    ir, nmr, tgt = case_inputs(case)
    g = load_golden(name)
    with torch.no_grad():
        mem = m.encoder(None, ir.to(DEV), None)
        logits = m(nmr.to(DEV), ir.to(DEV), tgt[:, :-1].to(DEV), None)
    if os.environ.get("MM_TEST_REPORT"):
        print(name, "bf16 memory", rel_err(mem[:2].float().cpu(), g["memory"]), "logits", rel_err(logits.cpu(), g["logits"]), flush=True)
    assert rel_err(mem[:2].float().cpu(), g["memory"]) < 2e-2
    assert rel_err(logits.cpu(), g["logits"]) < BF16_LOGITS_TOL[name]


@pytest.mark.parametrize("name", ["tiny_sm", "d100_sm", "d128_sdpa"])
def test_backward_fp32_vs_reference_golden(golden_meta, name):
    case = golden_meta["cases"][name]
    m = build(case)
    ir, nmr, tgt = case_inputs(case)
    g = load_golden(name)
    ir_d = ir.to(DEV).requires_grad_(True)
    tg = tgt.to(DEV)
    logits = m(nmr.to(DEV), ir_d, tg[:, :-1], None)
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.size(-1)), tg[:, 1:].reshape(-1), ignore_index=0)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert rel_err(ir_d.grad.cpu(), g["grad.ir"]) < 1e-3
    grads = dict((k, p.grad) for k, p in m.named_parameters())
    for k, v in g.items():
        if k.startswith("grad.") and k != "grad.ir":
            assert rel_err(grads[k[5:]].cpu(), v) < 1e-3, k
    with open(os.path.join(GOLDEN, f"{name}.gradsums.json")) as fh:
        sums = json.load(fh)
    for k, (s, a) in sums.items():
        got = float(grads[k].double().abs().sum())
        assert abs(got - a) <= 2e-3 * a + 1e-9, (k, got, a)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bucket_direct_weight_grads_equal_autograd(golden_meta, dtype, monkeypatch):
    """With dp.DataParallel the decoder node writes weight gradients straight into the (fresh) bucket slots, q/k/v
    stacked in one GEMM; the result must equal what autograd accumulates without buckets, a second backward must
    accumulate on top, and the optimizer step must invalidate the cached bf16 operands."""
    from multimodal_amd.dp import DataParallel
    from multimodal_amd.optim import OrthoGradAdamW
    case = golden_meta["cases"]["d100_sm"]
    ir, nmr, tgt = case_inputs(case)
    tg = tgt.to(DEV)

    def loss_of(m):
        logits = m(nmr.to(DEV), ir.to(DEV), tg[:, :-1], None)
        return torch.nn.functional.cross_entropy(logits.reshape(-1, logits.size(-1)), tg[:, 1:].reshape(-1), ignore_index=0)

    m = build(case, dtype)
    loss_of(m).backward()
    ref = dict((k, p.grad.clone()) for k, p in m.named_parameters())
    for p in m.parameters():
        p.grad = None
    dp = DataParallel(m)
    for b in dp.buckets.buckets:
        b["flat"].fill_(float("nan"))                                              # zero_grad must not rely on a memset
    lay = m.decoder.layers[0]
    assert lay.k.weight._mm_slot[1] == lay.q.weight._mm_slot[1] + lay.q.weight.numel()
    dp.zero_grad()
    loss_of(m).backward()
    assert not lay.q.weight._mm_fresh and not lay.ffn_w1.weight._mm_fresh          # the direct path ran
    assert lay.q.bias.grad is not lay.q.bias._mm_view                              # small gradients: gathered by finish()
    dp.finish()
    for k, p in m.named_parameters():
        assert p.grad is p._mm_view and torch.equal(p.grad, ref[k]), k
    loss_of(m).backward()                                                           # no zero_grad: accumulates
    for k, p in m.named_parameters():
        assert rel_err(p.grad, 2 * ref[k]) < 1e-6, k
    # optimizer step moves the masters; the next forward must see the new weights
    l0 = loss_of(m).item()
    opt = OrthoGradAdamW(m.parameters(), lr=1e-2)
    dp.zero_grad(); loss_of(m).backward(); dp.finish(); opt.step()
    l1 = loss_of(m).item()
    m2 = build(case, dtype)
    m2.load_state_dict(m.state_dict())
    assert l1 != l0 and abs(loss_of(m2).item() - l1) < 1e-6 * abs(l1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_optimizer_in_backward_equals_step_after_backward(golden_meta, dtype):
    """dp.attach_optimizer updates each gradient bucket on the side stream while backward continues; three steps must
    leave exactly the weights (and bf16 operand copies, seen through the next forward) of the plain sequence."""
    from multimodal_amd.dp import DataParallel
    from multimodal_amd.optim import OrthoGradAdamW
    case = golden_meta["cases"]["d100_sm"]
    ir, nmr, tgt = case_inputs(case)
    tg = tgt.to(DEV)

    def run(overlap):
        torch.manual_seed(7)                                       # DropPath draws
        m = build(case, dtype).train()
        dp = DataParallel(m, bucket_bytes=1 << 20)                 # several buckets even for this small model
        opt = OrthoGradAdamW(m.parameters(), lr=3e-3, weight_decay=0.01, ortho_eps=1e-8)
        if overlap:
            dp.attach_optimizer(opt)
        losses = []
        for _ in range(3):
            dp.zero_grad()
            logits = m(nmr.to(DEV), ir.to(DEV), tg[:, :-1], None)
            loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.size(-1)), tg[:, 1:].reshape(-1), ignore_index=0)
            loss.backward()
            dp.finish()
            opt.step()
            losses.append(loss.item())
        torch.cuda.synchronize()
        return losses, dict((k, v.clone()) for k, v in m.state_dict().items())

    l0, s0 = run(False)
    l1, s1 = run(True)
    assert l0 == l1
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_train_mode_droppath_vs_reference_golden(golden_meta, monkeypatch):
    """DropPath draws torch.rand((B,1,1)) per block in block order (convnext.py:15); the draws are taken from the
    CPU generator here so the stream equals the one the golden vector was produced with."""
    import multimodal_amd.models.convnext as cx
    case = golden_meta["cases"]["tiny_sm"]
    m = build(case).train()
    ir, nmr, tgt = case_inputs(case)
    g = load_golden("tiny_sm")
    real_rand = torch.rand

    def cpu_rand(shape, dtype=None, device=None):
        return real_rand(shape, dtype=dtype).to(device)
    monkeypatch.setattr(cx.torch, "rand", cpu_rand)
    torch.manual_seed(2024)
    with torch.no_grad():
        logits = m(nmr.to(DEV), ir.to(DEV), tgt[:, :-1].to(DEV), None)
    assert rel_err(logits.cpu(), g["logits_train_seed2024"]) < 1e-4


@pytest.mark.parametrize("name", ["tiny_sm", "d100_sm", "d128_sdpa", "test_config"])
def test_greedy_generate_ids_equal_reference(golden_meta, name):
    from multimodal_amd.generate import ids_to_lists
    case = golden_meta["cases"][name]
    m = build(case)
    gold = golden_meta["greedy"][name]
    ir8 = fill.synth_ir(8, 1000, 5).to(DEV)
    nmr8 = fill.synth_tokens(8, case["N"], 0, 1000, 5).to(DEV)
    toks = m.generate(nmr8, ir8, max_len=gold["max_len"])
    assert ids_to_lists(toks.cpu()) == gold["ids"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["tiny_sm", "d128_sdpa"])
def test_greedy_graph_replay_equals_eager_loop(golden_meta, name, dtype):
    """The decode step replayed from one captured HIP graph (device-side position / column / current ids) returns the ids
    of the eager loop, with and without early stop, and for a second call (a new capture) on another batch."""
    case = golden_meta["cases"][name]
    m = build(case, dtype)
    for seed, early in ((5, True), (6, False)):
        ir8 = fill.synth_ir(8, 1000, seed).to(DEV)
        nmr8 = fill.synth_tokens(8, case["N"], 0, 1000, seed).to(DEV)
        a = m.generate(nmr8, ir8, max_len=24, early_stop=early, graph=False)
        b = m.generate(nmr8, ir8, max_len=24, early_stop=early, graph=True)
        assert a.shape == b.shape and torch.equal(a, b)


def test_greedy_strings_equal_oracle(golden_meta):
    from multimodal_amd.generate import ids_to_strings
    from multimodal_amd.models import SmilesTokenizer
    case = golden_meta["cases"]["tiny_sm"]
    m = build(case)
    tok = SmilesTokenizer(os.path.join(GOLDEN, "vocab.txt"))
    ir8 = fill.synth_ir(8, 1000, 5)
    nmr8 = fill.synth_tokens(8, case["N"], 0, 1000, 5)
    toks = m.generate(nmr8.to(DEV), ir8.to(DEV), max_len=24)
    got = ids_to_strings(toks.cpu(), tok)
    cfg = cfg_of(case)
    seqs = restate.greedy_decode_ids(formula_state_dict(cfg), cfg, nmr8, ir8, 24)
    want = [restate.ids_to_string(s, tok.ids_to_tokens) for s in seqs]
    assert got == want


def test_block_api_bcl_layout():
    """ConvNeXtBlock1D.forward keeps the reference's [B,C,L] signature."""
    from multimodal_amd.models import ConvNeXtBlock1D
    blk = ConvNeXtBlock1D(128, drop_path=0.0, layer_scale_init_value=0.5)
    fill.fill_state_dict_(blk, seed=3)
    x = fill.synth_ir(2, 128 * 40, 1).view(2, 128, 40)
    sd = {"b." + k: v for k, v in blk.state_dict().items()}
    ref = restate.convnext_block(sd, "b.", x)
    y = blk.to(DEV).eval()(x.to(DEV))
    assert y.shape == x.shape
    assert rel_err(y.cpu(), ref) < 1e-4


def test_cross_attention_block_vs_text_restatement():
    """PARITY UNPINNED (reference class is dead code and cannot be imported): checked against oracle restatement."""
    from multimodal_amd.models import CrossAttentionBlock
    blk = CrossAttentionBlock(128, num_heads=4, dropout=0.0)
    fill.fill_state_dict_(blk, seed=4)
    q = fill.synth_ir(2, 9 * 128, 2).view(2, 9, 128) - 0.5
    kv = fill.synth_ir(2, 31 * 128, 3).view(2, 31, 128) - 0.5
    sd = {"x." + k: v for k, v in blk.state_dict().items()}
    ref = restate.cross_attention_block(sd, "x.", q, kv, 4)
    y = blk.to(DEV).eval()(q.to(DEV), kv.to(DEV))
    assert rel_err(y.cpu(), ref) < 1e-4
    # more queries than keys (the reference class accepts any Tq / Tkv): query chunks of at most Tkv rows
    q2 = fill.synth_ir(2, 70 * 128, 5).view(2, 70, 128) - 0.5
    ref2 = restate.cross_attention_block(sd, "x.", q2, kv, 4)
    y2 = blk(q2.to(DEV), kv.to(DEV))
    assert y2.shape == q2.shape and rel_err(y2.cpu(), ref2) < 1e-4
    # the class's own default constructor (dropout=0.1): the reference drops attention probabilities in every mode
    # (cross_attention.py:42), so the output is random; it must run, be reproducible under a seed, differ between seeds, and
    # its mean over many draws must approach the dropout-free block's pre-norm expectation closely enough to land near `y`
    blk_d = CrossAttentionBlock(128, num_heads=4)
    assert blk_d.dropout.p == 0.1
    blk_d.load_state_dict(blk.state_dict())
    blk_d = blk_d.to(DEV).eval()
    torch.manual_seed(7); a = blk_d(q.to(DEV), kv.to(DEV))
    torch.manual_seed(7); b = blk_d(q.to(DEV), kv.to(DEV))
    torch.manual_seed(8); c = blk_d(q.to(DEV), kv.to(DEV))
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.isfinite(a).all()
    mean = torch.stack([blk_d(q.to(DEV), kv.to(DEV)) for _ in range(400)]).mean(0)
    assert rel_err(mean.cpu(), y.cpu()) < 5e-2


# ---- the "own" GEMM backend: every dense product of the decoder on libmmhip (no library GEMM anywhere on the path)
@pytest.mark.parametrize("name", ["tiny_sm", "d100_sm", "d128_sdpa"])
def test_own_gemm_backend_fp32_forward_backward_vs_reference_golden(golden_meta, name):
    import multimodal_amd as mm
    case = golden_meta["cases"][name]
    m = build(case)
    mm.set_gemm_backend("own")
    ir, nmr, tgt = case_inputs(case)
    g = load_golden(name)
    ir_d = ir.to(DEV).requires_grad_(True)
    tg = tgt.to(DEV)
    logits = m(nmr.to(DEV), ir_d, tg[:, :-1], None)
    assert rel_err(logits.detach().cpu(), g["logits"]) < 1e-4 and elem_err(logits.detach().cpu(), g["logits"]) < 1e-4
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.size(-1)), tg[:, 1:].reshape(-1), ignore_index=0)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert rel_err(ir_d.grad.cpu(), g["grad.ir"]) < 1e-3
    grads = dict((k, p.grad) for k, p in m.named_parameters())
    for k, v in g.items():
        if k.startswith("grad.") and k != "grad.ir":
            assert rel_err(grads[k[5:]].cpu(), v) < 1e-3, k


@pytest.mark.parametrize("name", ["d100_sm", "d128_sdpa"])
def test_own_gemm_backend_bf16_matches_default_backend(golden_meta, name):
    """bf16: logits within the bf16 bar of the reference, and the gradients of the two backends within bf16 noise of each other."""
    import multimodal_amd as mm
    case = golden_meta["cases"][name]
    ir, nmr, tgt = case_inputs(case)
    g = load_golden(name)
    res = {}
    for backend in ("auto", "own"):
        m = build(case, torch.bfloat16)               # eval mode: DropPath is the identity, gradients still flow
        mm.set_gemm_backend(backend)
        tg = tgt.to(DEV)
        logits = m(nmr.to(DEV), ir.to(DEV), tg[:, :-1], None)
        assert rel_err(logits.detach().cpu(), g["logits"]) < 2e-2
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.size(-1)), tg[:, 1:].reshape(-1), ignore_index=0)
        loss.backward()
        res[backend] = {k: p.grad.float().cpu() for k, p in m.named_parameters() if p.grad is not None}
    worst = max(float((res["own"][k] - res["auto"][k]).norm() / res["auto"][k].norm().clamp_min(1e-20)) for k in res["auto"])
    assert worst < 6e-2, worst
