#!/usr/bin/env python3
"""Decoder-layer GEMMs at the config batch (10080 rows, E = 1600): library product + separate elementwise pass against the
large-tile kernel with that pass in its epilogue."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd  # noqa: E402,F401  (sets the library's tile mode)
from multimodal_amd import ops  # noqa: E402
from kernel_bench import timeit  # noqa: E402

d = torch.device("cuda:0")
M, E, F4 = 10080, 1600, 6400
bf = torch.bfloat16
r = lambda *s: (torch.rand(*s, device=d) * 2 - 1).to(bf)
xf, w1, b1 = r(M, E), r(F4, E) * 0.05, torch.randn(F4, device=d)
b1h = b1.to(bf)
t_lib = timeit(lambda: torch.addmm(b1h, xf, w1.t()))
h = torch.addmm(b1h, xf, w1.t())
t_relu = timeit(lambda: ops.relu2_fwd(h))
hh = torch.empty_like(h)
t_f = timeit(lambda: ops.gemm(xf, w1, M, F4, E, bias=b1, pre=hh, act=ops.ACT_RELU2))
t_p = timeit(lambda: ops.gemm(xf, w1, M, F4, E, bias=b1))
print(f"W1 fwd: lib addmm {t_lib*1e6:.1f} + relu2 {t_relu*1e6:.1f} = {(t_lib+t_relu)*1e6:.1f} us | big fused(pre+act) {t_f*1e6:.1f} us | big plain {t_p*1e6:.1f}", flush=True)

dout, w2t = r(M, E), r(F4, E) * 0.05          # dh = dout . W2, W2^T copy [F4][E]
t_lib = timeit(lambda: torch.mm(dout, w2t.t()))
dh = torch.mm(dout, w2t.t())
t_rb = timeit(lambda: ops.relu2_bwd_colsum_(h, dh))
t_f = timeit(lambda: ops.gemm(dout, w2t, M, F4, E, mul=h, act=ops.ACT_RELU2))
t_cs = timeit(lambda: ops.colsum(dh, M, F4))
print(f"dh bwd: lib mm {t_lib*1e6:.1f} + relu2_bwd_colsum {t_rb*1e6:.1f} = {(t_lib+t_rb)*1e6:.1f} us | big fused(mul) {t_f*1e6:.1f} + colsum {t_cs*1e6:.1f}", flush=True)

xa, wq, bq = r(M, E), r(3 * E, E) * 0.05, torch.randn(3 * E, device=d)
bqh = bq.to(bf)
t_lib = timeit(lambda: torch.addmm(bqh, xa, wq.t()))
t_b = timeit(lambda: ops.gemm(xa, wq, M, 3 * E, E, bias=bq))
print(f"QKV fwd: lib addmm {t_lib*1e6:.1f} | big {t_b*1e6:.1f}", flush=True)
# d_o = dx1 . Wo (NN on [E][E]) vs NT
dx1, wo = r(M, E), r(E, E) * 0.05
t_nn = timeit(lambda: torch.mm(dx1, wo))
t_nt = timeit(lambda: ops.gemm(dx1, wo, M, E, E))
print(f"d_o: lib NN {t_nn*1e6:.1f} | big NT {t_nt*1e6:.1f}", flush=True)
# weight gradients (f32 out): dW1 = dh^T xf etc.
for (n, k, nm) in ((F4, E, "dW1"), (E, F4, "dW2"), (3 * E, E, "dWqkv"), (E, E, "dWo")):
    dy, x = r(M, n), r(M, k)
    t = timeit(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32))
    print(f"{nm}: lib TN f32 {t*1e6:.1f} us = {2*M*n*k/t/1e15:.2f} PF", flush=True)
