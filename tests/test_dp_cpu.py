"""N>1 path on CPU: 2-rank gloo step == 1-rank step on the concatenated batch (SURVEY.md 8e equivalence target).
The DataParallel wrapper is model-agnostic; a small torch module stands in for the GPU-only hot path here."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.LayerNorm(32), torch.nn.Linear(32, 5))


def _data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(8, 16, generator=g), torch.randint(0, 5, (8,), generator=g)


def _worker(rank, world, port, q, prescale=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_amd.dp import DataParallel
    m = _model()
    if rank == 1:                                  # replicas must be made identical by the wrapper
        for p in m.parameters():
            p.data.add_(1.0)
    dp = DataParallel(m, bucket_bytes=1 << 10)     # tiny buckets: several collectives, overlap path exercised
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    x, y = _data()
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for _ in range(2):
        dp.zero_grad()
        loss = torch.nn.functional.cross_entropy(dp(xs), ys)
        (dp.scale_loss(loss) if prescale else loss).backward()
        dp.finish()
        opt.step()
    if rank == 0:
        q.put([p.detach().numpy().tolist() for p in m.parameters()])
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("prescale", [False, True])
def test_two_rank_step_equals_single_rank_on_concatenated_batch(prescale):
    """prescale: dp.scale_loss() divides the loss by the world size, finish() then skips its division pass."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000 + (1000 if prescale else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, prescale)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m = _model()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    x, y = _data()
    for _ in range(2):
        opt.zero_grad()
        torch.nn.functional.cross_entropy(m(x), y).backward()
        opt.step()
    for a, b in zip(got, m.parameters()):
        assert torch.allclose(torch.tensor(a), b.detach(), atol=1e-6, rtol=1e-5)


def test_grad_buckets_are_views():
    from multimodal_amd.dp import GradBuckets
    m = _model()
    gb = GradBuckets(list(m.parameters()), bucket_bytes=1 << 9)
    assert len(gb.buckets) >= 2
    x, y = _data()
    torch.nn.functional.cross_entropy(m(x), y).backward()
    for b in gb.buckets:
        off = 0
        for p in b["params"]:
            assert p.grad.data_ptr() == b["flat"].data_ptr() + off * 4
            assert p._mm_slot[0] is b["flat"] and p._mm_slot[1] == off
            off += -(-p.numel() // GradBuckets.ALIGN) * GradBuckets.ALIGN      # slots are 256-byte aligned
        assert b["flat"].abs().sum() > 0


def test_grad_buckets_keep_groups_adjacent():
    from multimodal_amd.dp import GradBuckets
    ps = [torch.nn.Parameter(torch.zeros(5, 3)) for _ in range(6)]
    gb = GradBuckets(ps, bucket_bytes=1 << 30, adjacent=[(ps[1], ps[3], ps[4])])
    assert sorted(id(p) for b in gb.buckets for p in b["params"]) == sorted(id(p) for p in ps)
    step = -(-15 // GradBuckets.ALIGN) * GradBuckets.ALIGN
    assert ps[3]._mm_slot[1] == ps[1]._mm_slot[1] + step and ps[4]._mm_slot[1] == ps[3]._mm_slot[1] + step
