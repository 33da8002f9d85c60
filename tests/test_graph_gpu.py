"""The train step captured in a HIP graph (multimodal_amd.graph.CapturedTrainStep) against the eager step: same model, same
batches, a learning rate that changes every step (device-resident optimizer scalars), parameters compared after 4 replays.
Also: OrthoGradAdamW(capturable=True) run eagerly equals the by-value form, and matches the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, case_inputs, rel_err
from oracle import fill

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _reset_dtype():
    import multimodal_amd as mm
    yield
    mm.set_compute_dtype(torch.float32)


def _model(case, dtype):
    import multimodal_amd as mm
    from multimodal_amd.models import MultiModalToSMILESModel
    m = MultiModalToSMILESModel(591, 1000, max_seq_length=case["max_seq"], max_nmr_length=case["max_nmr"],
                                max_memory_length=case["max_mem"], embed_dim=case["E"], num_heads=case["H"],
                                num_layers=case["L"], dropout=0.0, verbose=False, use_stablemax=case["sm"])
    fill.fill_state_dict_(m, seed=0)
    mm.set_compute_dtype(dtype)
    return m.to(DEV).eval()            # eval: no DropPath draws, so eager and replay see the same arithmetic


LRS = [3e-3, 1e-3, 2e-3, 5e-4, 4e-3, 1e-3, 2.5e-3]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_captured_step_equals_eager_step(golden_meta, dtype):
    from multimodal_amd.dp import DataParallel
    from multimodal_amd.graph import CapturedTrainStep
    from multimodal_amd.optim import OrthoGradAdamW
    case = golden_meta["cases"]["d100_sm"]
    ir, nmr, tgt = (t.to(DEV) for t in case_inputs(case))
    batches = [(ir * (1.0 - 0.1 * i), nmr, tgt.roll(i, 0)) for i in range(len(LRS))]

    def run(captured):
        m = _model(case, dtype)
        dp = DataParallel(m, bucket_bytes=1 << 20)
        opt = OrthoGradAdamW(m.parameters(), lr=LRS[0], weight_decay=0.01, eps=1e-8, ortho_eps=1e-8, capturable=True)
        s_ir, s_nmr, s_tgt = (t.clone() for t in batches[0])
        losses = []

        def step():
            dp.zero_grad()
            lg = m(s_nmr, s_ir, s_tgt[:, :-1], None)
            loss = torch.nn.functional.cross_entropy(lg.reshape(-1, lg.size(-1)), s_tgt[:, 1:].reshape(-1), ignore_index=0)
            loss.backward()
            dp.finish()
            opt.step()
            return loss.detach()

        def feed(i):
            for g in opt.param_groups:
                g["lr"] = LRS[i]
            s_ir.copy_(batches[i][0]); s_nmr.copy_(batches[i][1]); s_tgt.copy_(batches[i][2])

        n_warm = 3
        if captured:
            cap = CapturedTrainStep(step, opt, warmup=n_warm, before_warmup_step=feed)   # real steps on batches 0..2
            for i in range(n_warm, len(LRS)):
                feed(i)
                losses.append(float(cap()))
        else:
            for i in range(len(LRS)):
                feed(i)
                l = step()
                if i >= n_warm:
                    losses.append(float(l))
        torch.cuda.synchronize()
        return losses, {k: v.detach().clone() for k, v in m.named_parameters()}, opt._step

    l_e, p_e, n_e = run(False)
    l_c, p_c, n_c = run(True)
    assert n_e == n_c == len(LRS)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for a, b in zip(l_e, l_c):
        assert abs(a - b) <= tol * abs(a), (l_e, l_c)
    for k in p_e:
        assert rel_err(p_c[k], p_e[k]) < tol, k


def test_capturable_optimizer_vs_reference_golden():
    from multimodal_amd.optim import OrthoGradAdamW
    from oracle.gen_golden_optim import NO_GRAD, STEPS, VARIANTS, params_and_grads
    gold = np.load(os.path.join(GOLDEN, "optim.npz"), allow_pickle=False)
    eps, rescale, lr, betas, wd = VARIANTS["cfg"]
    ps0, gs = params_and_grads()
    ps = [torch.nn.Parameter(p.clone().to(DEV)) for p in ps0]
    opt = OrthoGradAdamW(ps, lr=lr, betas=betas, eps=eps, weight_decay=wd, ortho_eps=eps, rescale=rescale, capturable=True)
    prev = [p.clone() for p in ps0]
    for t in range(STEPS):
        for p, g in zip(ps, gs[t]):
            p.grad = None if g is None else g.clone().to(DEV)
        opt.step()
        torch.cuda.synchronize()
        for i, p in enumerate(ps):
            if i == NO_GRAD:
                continue
            want = torch.from_numpy(gold[f"cfg.step{t}.p{i}"])
            err = (p.detach().cpu() - want).double()
            assert float(err.abs().max()) <= 0.05 * lr and float(err.norm()) <= 1e-4 * float((want - prev[i]).double().norm()) + 1e-9
            prev[i] = want
            with torch.no_grad():
                p.copy_(want)
