import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


ACHIEVED = []          # (test, text): achieved errors of the parity tests, printed in the terminal summary of every run


def report(*parts):
    """Record achieved errors for the run's log (pytest hides a passing test's stdout): shown under 'achieved errors'."""
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0].split("::", 1)[-1]
    ACHIEVED.append((test, " ".join(f"{x:.3g}" if isinstance(x, float) else str(x) for x in parts)))


def pytest_terminal_summary(terminalreporter):
    if ACHIEVED:
        terminalreporter.section("achieved errors (parity tests)")
        for test, text in ACHIEVED:
            terminalreporter.write_line(f"{test}: {text}")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(GOLDEN, "meta.json")) as fh:
        return json.load(fh)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))


def cfg_of(case):
    from oracle.restate import Cfg
    return Cfg(embed_dim=case["E"], num_heads=case["H"], num_layers=case["L"], max_seq_length=case["max_seq"],
               max_nmr_length=case["max_nmr"], max_memory_length=case["max_mem"], use_stablemax=case["sm"])


def case_inputs(case, seed=0):
    from oracle import fill
    ir = fill.synth_ir(case["B"], 1000, seed)
    nmr = fill.synth_tokens(case["B"], case["N"], 0, 1000, seed)
    tgt = fill.synth_tokens(case["B"], case["T"] + 1, 15, 591, seed + 1)
    tgt[:, 0] = 12
    return ir, nmr, tgt


def rel_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def elem_err(a, b):
    """Elementwise companion of rel_err (a max-norm measure lets an element near zero be 100 % wrong): the largest
    |a - b| / (|b| + rms(b)) over all elements.  `elem_err(a, b) < tol` is `|a-b| <= tol * (|b| + rms(b))` everywhere."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    rms = b.pow(2).mean().sqrt().clamp_min(1e-30)
    return float(((a - b).abs() / (b.abs() + rms)).max())


@pytest.fixture(scope="session")
def config_meta():
    with open(os.path.join(GOLDEN, "meta_config.json")) as fh:
        return json.load(fh)


def l2_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def check_grads_against_golden(grads, g, tol, name="", l2_tol=None):
    """grads: {parameter name: gradient tensor}; g: a golden npz dict with 'grad.<name>' entries (nmr_embed by rows).
    tol bounds the max-norm error of every tensor, l2_tol (optional) its relative L2 error."""
    n = 0
    for k, v in g.items():
        if not k.startswith("grad.") or k == "grad.ir" or k.endswith(".rows"):
            continue
        got = grads[k[5:]].detach().float().cpu()
        if k + ".rows" in g:
            got = got[torch.from_numpy(g[k + ".rows"])]
        assert rel_err(got, v) < tol, (name, k, rel_err(got, v))
        if l2_tol is not None:
            assert l2_err(got, v) < l2_tol, (name, k, l2_err(got, v))
        n += 1
    return n
