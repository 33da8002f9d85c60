"""N>1 rehearsal on ONE GPU: two processes share cuda:0 and exchange gradients over gloo (RCCL refuses two ranks on
one device; the 8-GPU RCCL run is the driver's).  Everything else is the production path: the real model on the HIP
kernels, gradient buckets, the hook-driven launch of gather + all-reduce on the side stream, finish().
Equivalence target: averaged 2-rank gradients == single-process gradients on the concatenated batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, case_inputs, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASE = dict(B=4, E=200, H=2, L=2, N=7, T=9, max_mem=32, max_nmr=190, max_seq=95, sm=True)


def _build(dtype):
    import multimodal_amd as mm
    from multimodal_amd.models import MultiModalToSMILESModel
    from oracle import fill
    c = CASE
    m = MultiModalToSMILESModel(591, 1000, max_seq_length=c["max_seq"], max_nmr_length=c["max_nmr"],
                                max_memory_length=c["max_mem"], embed_dim=c["E"], num_heads=c["H"], num_layers=c["L"],
                                dropout=0.0, verbose=False, use_stablemax=c["sm"])
    fill.fill_state_dict_(m, seed=0)
    mm.set_compute_dtype(dtype)
    return m.to(DEV).eval()


def _loss(m, ir, nmr, tgt, denom):
    tg = tgt.to(DEV)
    logits = m(nmr.to(DEV), ir.to(DEV), tg[:, :-1], None)
    ce = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.size(-1)), tg[:, 1:].reshape(-1), ignore_index=0,
                                           reduction="sum")
    return ce / denom


def _worker(rank, world, port, dtype, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_amd.dp import DataParallel
    m = _build(dtype)
    if rank == 1:                                   # the wrapper must make the replicas identical
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.5)
    dp = DataParallel(m, bucket_bytes=1 << 20)      # several buckets: several collectives launched from hooks
    assert dp.overlap and len(dp.buckets.buckets) > 2
    ir, nmr, tgt = case_inputs(CASE)
    n_tok = int((tgt[:, 1:] != 0).sum())
    h = CASE["B"] // world
    sl = slice(rank * h, (rank + 1) * h)
    out = None
    for _ in range(2):                              # second step: zero_grad() state reset, no stale gradients
        dp.zero_grad()
        # sum over this rank's tokens / global token count, x world: the rank average is the global-batch mean
        dp.scale_loss(_loss(m, ir[sl], nmr[sl], tgt[sl], n_tok / world)).backward()     # finish() skips its division
        launched = sum(bool(b.get("launched")) for b in dp.buckets.buckets)
        dp.finish()
        torch.cuda.synchronize()
        out = [p.grad.detach().float().cpu().numpy() for p in m.parameters()]
    if rank == 0:
        q.put((launched, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_rank_gradients_equal_single_rank_on_concatenated_batch(dtype):
    import multimodal_amd as mm
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 1000 + (0 if dtype == torch.float32 else 1)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, dtype, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        launched, got = q.get(timeout=240)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs)
    assert launched > 0                             # buckets were reduced from inside backward, not only by finish()
    try:
        m = _build(dtype)
        ir, nmr, tgt = case_inputs(CASE)
        n_tok = int((tgt[:, 1:] != 0).sum())
        _loss(m, ir, nmr, tgt, n_tok).backward()
        tol = 2e-5 if dtype == torch.float32 else 3e-2    # bf16: the two halves round their GEMM operands separately
        for (k, p), g in zip(m.named_parameters(), got):
            assert rel_err(torch.from_numpy(g), p.grad.float().cpu()) < tol, k
    finally:
        mm.set_compute_dtype(torch.float32)


def _rccl_worker(port, q):
    """One rank on the nccl (= RCCL) backend with every collective forced: communicator creation, broadcast, the
    hook-driven asynchronous bucket all-reduce on RCCL's stream, finish(), accumulate() and the second-backward guard."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device(DEV)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    res = {}
    try:
        from multimodal_amd.dp import DataParallel
        m = _build(torch.float32)
        dp = DataParallel(m, bucket_bytes=1 << 20, force_collectives=True)
        assert dp.collective and dp.overlap and not dp.host_staged and dist.get_backend() == "nccl"
        ir, nmr, tgt = case_inputs(CASE)
        n_tok = int((tgt[:, 1:] != 0).sum())
        for _ in range(2):
            dp.zero_grad()
            dp.scale_loss(_loss(m, ir, nmr, tgt, n_tok)).backward()
            launched = sum(bool(b.get("launched")) for b in dp.buckets.buckets)
            dp.finish()
        torch.cuda.synchronize()
        res["launched"] = launched
        res["n_allreduce"] = dp.n_allreduce
        res["n_buckets"] = len(dp.buckets.buckets)
        res["grads"] = [p.grad.detach().float().cpu().numpy() for p in m.parameters()]
        # a second backward without zero_grad() would add to reduced buckets: refused, not silently wrong
        try:
            _loss(m, ir, nmr, tgt, n_tok).backward()
            res["second_backward"] = "accepted"
        except RuntimeError as e:
            res["second_backward"] = str(e)
        # gradient accumulation: first micro-batch inside accumulate(), ONE all-reduce per bucket at the end
        dp.zero_grad()
        n0 = dp.n_allreduce
        h = CASE["B"] // 2
        with dp.accumulate():
            _loss(m, ir[:h], nmr[:h], tgt[:h], n_tok).backward()
        assert dp.n_allreduce == n0
        _loss(m, ir[h:], nmr[h:], tgt[h:], n_tok).backward()
        dp.finish()
        torch.cuda.synchronize()
        res["accum_allreduce"] = dp.n_allreduce - n0
        res["accum_grads"] = [p.grad.detach().float().cpu().numpy() for p in m.parameters()]
    finally:
        q.put(res)
        dist.destroy_process_group()


def test_rccl_backend_one_rank_runs_the_exchange_path():
    """The nccl backend IS RCCL on ROCm.  A one-GPU box cannot host a second rank on the card, so the product's exchange
    path runs here with one rank and forced collectives: results must equal the plain single-process gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29800 + os.getpid() % 1000, q))
    p.start()
    try:
        res = q.get(timeout=240)
    finally:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    assert p.exitcode == 0 and "grads" in res, res.keys()
    assert res["launched"] > 0 and res["n_allreduce"] == 2 * res["n_buckets"] and res["n_buckets"] > 2
    assert "accumulate" in res["second_backward"]
    assert res["accum_allreduce"] == res["n_buckets"]
    m = _build(torch.float32)
    ir, nmr, tgt = case_inputs(CASE)
    n_tok = int((tgt[:, 1:] != 0).sum())
    _loss(m, ir, nmr, tgt, n_tok).backward()
    for (k, prm), g, ga in zip(m.named_parameters(), res["grads"], res["accum_grads"]):
        assert rel_err(torch.from_numpy(g), prm.grad.float().cpu()) < 1e-6, k
        assert rel_err(torch.from_numpy(ga), prm.grad.float().cpu()) < 2e-5, k       # two half batches: other sum order
