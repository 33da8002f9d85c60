#!/usr/bin/env python3
"""Greedy decode at the nocat shape (B=1024, 80 steps) and its per-step GEMM shapes, libmmhip vs library:
    python benchmarks/decode_bench.py            # MM_DECODE_GEMM=mm|lib selects the decode-step products"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd as mm  # noqa: E402
from multimodal_amd import ops  # noqa: E402
from bench import CONFIGS, build_model, synth_batch  # noqa: E402
from kernel_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
if "gemm" in sys.argv:
    for (M, N, K) in [(1024, 3072, 1024), (1024, 1024, 1024), (1024, 4096, 1024), (1024, 1024, 4096), (1024, 591, 1024),
                      (128, 3072, 1024), (128, 4096, 1024), (128, 1024, 4096)]:
        a = (torch.rand(M, K, device=dev) - 0.5).to(torch.bfloat16)
        w = (torch.rand(N, K, device=dev) - 0.5).to(torch.bfloat16)
        b = torch.zeros(N, device=dev)
        bb = b.to(torch.bfloat16)
        res = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        t1 = timeit(lambda: ops.gemm(a, w, M, N, K, bias=b, res=res), iters=50)
        t2 = timeit(lambda: torch.addmm(bb, a, w.t()).add_(res), iters=50)
        t3 = timeit(lambda: torch.addmm(bb, a, w.t()), iters=50)
        print(f"gemm M{M} N{N} K{K}: mm+bias+res {t1 * 1e6:6.1f} us | addmm+add {t2 * 1e6:6.1f} us | addmm alone {t3 * 1e6:6.1f} us", flush=True)
c = CONFIGS["local_config_nocat"]
for dt in (torch.bfloat16, torch.float32):
    B = 1024 if dt == torch.bfloat16 else 256
    m = build_model(c, dev).eval()
    ir, nmr, _ = synth_batch(c, B, dev, 4321)
    with mm.compute_dtype(dt):
        m.generate(nmr, ir, max_len=80, early_stop=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks = m.generate(nmr, ir, max_len=80, early_stop=False)
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t0
    print(json.dumps({"dtype": str(dt), "B": B, "tokens_per_s": round(B * 80 / dtm, 1), "seconds": round(dtm, 4),
                      "decode_gemm": os.environ.get("MM_DECODE_GEMM", "mm")}), flush=True)
