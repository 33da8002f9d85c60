#!/usr/bin/env python3
"""NT bf16 products with about 1000 rows (encoder stage 3 at the config batch, decode steps): every large-tile shape
(MM_GEMM_BIG_SHAPE is read once per process, so each shape runs in a child process) vs gemm_k vs the library."""
import os
import subprocess
import sys

SHAPES = [(992, 6400, 1600), (992, 1600, 6400), (992, 1600, 1600), (992, 1600, 512), (1024, 3072, 1024), (1024, 4096, 1024),
          (1024, 1024, 4096), (1024, 1024, 1024), (1984, 1024, 256), (1984, 256, 1024)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from multimodal_amd import ops
    from kernel_bench import timeit
    dev = torch.device("cuda:0")
    out = []
    for (M, N, K) in SHAPES:
        a = (torch.rand(M, K, device=dev) - 0.5).to(torch.bfloat16)
        w = (torch.rand(N, K, device=dev) - 0.5).to(torch.bfloat16)
        b = torch.zeros(N, device=dev)
        res = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        if sys.argv[2] == "lib":
            bb = b.to(torch.bfloat16)
            t = timeit(lambda: torch.addmm(bb, a, w.t()).add_(res), iters=30)
        else:
            y = ops.gemm(a, w, M, N, K, bias=b, res=res)
            ref = torch.addmm(b.to(torch.bfloat16), a, w.t()).add_(res)
            err = float((y.float() - ref.float()).abs().max() / ref.float().abs().max())
            assert err < 2e-2, (M, N, K, err)
            t = timeit(lambda: ops.gemm(a, w, M, N, K, bias=b, res=res), iters=30)
        out.append(f"{t * 1e6:6.1f}")
    print(sys.argv[2].ljust(8), " ".join(out), flush=True)
    sys.exit(0)
print("shape   ", " ".join(f"{m}x{n}x{k}" for m, n, k in SHAPES))
for tag, env in [("lib", {}), ("gemm_k", {"MM_GEMM_NO256": "1"}), ("auto", {})] + [(f"big{i}", {"MM_GEMM_BIG_SHAPE": str(i), "MM_GEMM_BIG_MIN_MN": "1"}) for i in (4, 5)] + [(f"b{i}s{sp}", {"MM_GEMM_BIG_SHAPE": str(i), "MM_GEMM_BIG_MIN_MN": "1", "MM_GEMM_BIG_SPLITS": str(sp)}) for i in (4, 5) for sp in (2, 4)]:
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", tag], env=e, check=False)
