"""Stand-alone timing of the library weight-gradient GEMMs of one decoder layer (dW = dY^T X, f32 out)."""
import torch

dev = "cuda:0"
M = 10080


def tm(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for (no, ni) in [(4800, 1600), (1600, 1600), (6400, 1600), (1600, 6400)]:
    dy = (torch.rand(M, no, device=dev) - 0.5).to(torch.bfloat16)
    x = (torch.rand(M, ni, device=dev) - 0.5).to(torch.bfloat16)
    g = torch.empty(no, ni, device=dev)
    gb = torch.empty(no, ni, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * no * ni
    t1 = tm(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32, out=g))
    t2 = tm(lambda: torch.mm(dy.t(), x, out=gb))
    dyt = dy.t().contiguous()
    xt = x.t().contiguous()
    t3 = tm(lambda: torch.mm(dyt, xt.t(), out_dtype=torch.float32, out=g))     # NT form of the same product
    print(f"dW {no}x{ni} K={M}: TN f32-out {t1:7.1f} us {fl / t1 / 1e9:6.3f} PF | TN bf16-out {t2:7.1f} us {fl / t2 / 1e9:6.3f} PF"
          f" | NT f32-out {t3:7.1f} us {fl / t3 / 1e9:6.3f} PF", flush=True)
