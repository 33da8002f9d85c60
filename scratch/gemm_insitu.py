import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops
d = torch.device("cuda:0")
M, N, K = 10080, 1600, 6400
a = (torch.rand(M, K, device=d) * 2 - 1).to(torch.bfloat16)
w = (torch.rand(N, K, device=d) * 2 - 1).to(torch.bfloat16)
bias = torch.randn(N, device=d); res = torch.randn(M, N, device=d).to(torch.bfloat16)
flush = torch.empty(1 << 28, dtype=torch.float32, device=d)   # 1 GiB
def run(fn, flush_between):
    ts = []
    for i in range(12):
        if flush_between: flush.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(s.elapsed_time(e) * 1e3)
    return sum(ts) / len(ts)
for name, fn in [("plain", lambda: ops.gemm(a, w, M, N, K)), ("bias", lambda: ops.gemm(a, w, M, N, K, bias=bias)),
                 ("bias+res", lambda: ops.gemm(a, w, M, N, K, bias=bias, res=res)),
                 ("torch addmm", lambda: torch.addmm(bias.to(torch.bfloat16), a, w.t()))]:
    print(name, "warm %.1f us" % run(fn, False), "cold %.1f us" % run(fn, True))
