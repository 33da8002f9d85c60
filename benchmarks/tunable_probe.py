#!/usr/bin/env python3
"""Library products of the decoder at the config batch with the default heuristic and with PyTorch TunableOp's pick."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd  # noqa: F401  (sets TENSILE_STREAMK_DATA_PARALLEL)
from kernel_bench import timeit  # noqa: E402
dev = torch.device("cuda:0")
M, E, F4 = 10080, 1600, 6400
x = torch.randn(M, E, device=dev, dtype=torch.bfloat16)
h = torch.randn(M, F4, device=dev, dtype=torch.bfloat16)
wq = torch.randn(3 * E, E, device=dev, dtype=torch.bfloat16); bq = torch.randn(3 * E, device=dev, dtype=torch.bfloat16)
w1 = torch.randn(F4, E, device=dev, dtype=torch.bfloat16); b1 = torch.randn(F4, device=dev, dtype=torch.bfloat16)
w2t = torch.randn(F4, E, device=dev, dtype=torch.bfloat16)
dy = torch.randn(M, E, device=dev, dtype=torch.bfloat16)
cases = {
    "qkv fwd addmm": lambda: torch.addmm(bq, x, wq.t()),
    "w1 fwd addmm": lambda: torch.addmm(b1, x, w1.t()),
    "dh = dy . W2 (NT on the transposed copy)": lambda: torch.mm(dy, w2t.t()),
    "dW1 = dh^T . x (f32 out)": lambda: torch.mm(h.t(), x, out_dtype=torch.float32),
    "dW2 = dy^T . a (f32 out)": lambda: torch.mm(dy.t(), h, out_dtype=torch.float32),
    "dWqkv (f32 out)": lambda: torch.mm(torch.randn(M, 3 * E, device=dev, dtype=torch.bfloat16).t(), x, out_dtype=torch.float32) if False else None,
}
del cases["dWqkv (f32 out)"]
base = {k: timeit(f, iters=30) for k, f in cases.items()}
import torch.cuda.tunable as T
T.enable(True); T.tuning_enable(True); T.set_max_tuning_duration(50); T.set_max_tuning_iterations(30)
T.set_filename("/tmp/tunable_probe.csv")
for f in cases.values():
    f()
torch.cuda.synchronize()
T.tuning_enable(False)
tuned = {k: timeit(f, iters=30) for k, f in cases.items()}
for k in cases:
    print(f"{k:44s} default {base[k]*1e6:7.1f} us   tuned {tuned[k]*1e6:7.1f} us", flush=True)
print(T.get_results())
