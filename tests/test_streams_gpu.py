"""Regression for the two-stream GEMM deadlock: a backward whose weight-gradient GEMMs run on the side
stream at B*S = 20160 rows -- the range where hipBLASLt's stream-K splitting made the main-stream and side-stream GEMMs wait
on each other forever -- must complete.  Runs in a child process under a time limit, so a regression fails instead of hanging
the test session."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys, torch
sys.path.insert(0, %r)
import multimodal_amd as mm
from multimodal_amd import functional as FN
from multimodal_amd.dp import DataParallel
from multimodal_amd.models import MultiModalToSMILESModel
from multimodal_amd.optim import OrthoGradAdamW
assert os.environ.get("TENSILE_STREAMK_DATA_PARALLEL") == "1"          # set by the package import
dev = torch.device("cuda", 0)
mm.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
m = MultiModalToSMILESModel(591, 1000, max_seq_length=95, max_nmr_length=190, max_memory_length=32, embed_dim=1600,
                            num_heads=16, num_layers=2, dropout=0.0, verbose=False, use_stablemax=True).to(dev).train()
B = 64
assert B * 315 > 16380 and B * 315 <= FN._defer_rows_limit() and m.decoder.layers[0].defer_param_grads
dp = DataParallel(m)
opt = OrthoGradAdamW(m.parameters(), lr=1e-4, eps=1e-8, ortho_eps=1e-8)
g = torch.Generator().manual_seed(1)
ir = torch.rand(B, 1000, generator=g).to(dev)
nmr = torch.randint(0, 1000, (B, 190), generator=g).to(dev)
tgt = torch.randint(15, 591, (B, 95), generator=g).to(dev)
for _ in range(4):
    dp.zero_grad()
    lg = m(nmr, ir, tgt[:, :-1], None)
    torch.nn.functional.cross_entropy(lg.reshape(-1, 591), tgt[:, 1:].reshape(-1)).backward()
    dp.finish()
    opt.step()
torch.cuda.synchronize()
print("completed", flush=True)
""" % ROOT


def test_two_stream_backward_completes_at_large_row_counts():
    env = dict(os.environ)
    env.pop("TENSILE_STREAMK_DATA_PARALLEL", None)
    env.pop("MM_DEFER_MAX_ROWS", None)
    try:
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=120)
    except subprocess.TimeoutExpired:
        pytest.fail("train step with side-stream weight gradients at 20160 rows did not finish in 120 s (GEMM deadlock?)")
    assert r.returncode == 0 and "completed" in r.stdout, r.stderr[-2000:]


def test_side_stream_runs_beside_the_main_stream_even_after_many_streams_exist():
    """HIP maps streams onto a few hardware queues in creation order; a side stream that lands on the main stream's queue is
    serialised with it (what happened once RCCL's streams existed: DESIGN.md 5b).  side_stream() must hand out a stream that
    really runs beside the current one."""
    import torch
    from multimodal_amd import functional as FN
    dev = torch.device("cuda", 0)
    crowd = [torch.cuda.Stream(device=dev) for _ in range(7)]          # shift the creation order as a process group does
    for s in crowd:
        with torch.cuda.stream(s):
            torch.zeros(1, device=dev)
    saved = FN._side_streams.pop(dev, None)
    try:
        side = FN.side_stream(dev)
        assert FN._runs_beside(torch.cuda.current_stream(dev), side)
    finally:
        if saved is not None:
            FN._side_streams[dev] = saved
