import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops
from bench import _event_time
d = torch.device("cuda", 0)
M, N, D = 10080, 6400, 1600
h = torch.randn(M, N, device=d).to(torch.bfloat16)
da = torch.randn(M, N, device=d).to(torch.bfloat16)
t1 = _event_time(lambda: ops.relu2_bwd_(h, da))
t2 = _event_time(lambda: ops.colsum(da, M, N))
t3 = _event_time(lambda: ops.relu2_bwd_colsum_(h, da))
print(f"relu2_bwd {t1*1e6:.1f} us, colsum {t2*1e6:.1f} us, fused {t3*1e6:.1f} us (bytes 387 MB -> {387e6/t3/1e12:.2f} TB/s)", flush=True)
x = torch.randn(M, D, device=d).to(torch.bfloat16)
w = torch.ones(D, device=d); b = torch.zeros(D, device=d)
_, mean, rstd = ops.layernorm_fwd(x, w, b, save_stats=True)
dy = torch.randn(M, D, device=d).to(torch.bfloat16)
dr = torch.randn(M, D, device=d).to(torch.bfloat16)
t4 = _event_time(lambda: ops.layernorm_bwd(x, dy, dr, w, mean, rstd))
t5 = _event_time(lambda: ops.layernorm_bwd_sums(x, dy, dr, w, mean, rstd))
t6 = _event_time(lambda: ops.colsum(dy, M, D))
print(f"ln_bwd {t4*1e6:.1f} us, ln_bwd_sums {t5*1e6:.1f} us, colsum[M,1600] {t6*1e6:.1f} us", flush=True)
