"""Stage-by-stage trace of the 2-rank one-GPU rehearsal (gloo on CUDA tensors)."""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T0 = time.time()


def say(rank, *a):
    print(f"[{time.time() - T0:6.1f}s r{rank}]", *a, flush=True)


def worker(rank, world, port):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    say(rank, "init")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    say(rank, "pg up")
    x = torch.full((1000,), float(rank + 1), device="cuda:0")
    dist.broadcast(x, 0)
    torch.cuda.synchronize()
    say(rank, "broadcast ok", x[0].item())
    y = torch.full((1000,), float(rank + 1), device="cuda:0")
    w = dist.all_reduce(y, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    say(rank, "all_reduce ok", y[0].item())
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        z = torch.full((1000,), float(rank + 1), device="cuda:0")
        w = dist.all_reduce(z, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    say(rank, "side all_reduce ok", z[0].item())
    import test_dp_gpu as T
    from multimodal_amd.dp import DataParallel
    m = T._build(torch.float32)
    say(rank, "model built")
    dp = DataParallel(m, bucket_bytes=1 << 20)
    say(rank, "dp wrapped", len(dp.buckets.buckets))
    from conftest import case_inputs
    ir, nmr, tgt = case_inputs(T.CASE)
    sl = slice(rank * 2, rank * 2 + 2)
    dp.zero_grad()
    loss = T._loss(m, ir[sl], nmr[sl], tgt[sl], 10.0)
    say(rank, "fwd", loss.item())
    loss.backward()
    say(rank, "bwd issued")
    dp.finish()
    torch.cuda.synchronize()
    say(rank, "finish ok")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, 29655)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(timeout=45)
    for p in ps:
        if p.is_alive():
            print("still alive: killing", p.pid, flush=True)
            p.kill()
    print("exit codes", [p.exitcode for p in ps], flush=True)
