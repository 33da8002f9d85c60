"""Does hipBLASLt read TENSILE_STREAMK_DATA_PARALLEL per call (set AFTER torch ran a GEMM) or once at start-up?
Prints the library GEMM grid sizes seen by a stream-K-prone product before and after setting the variable."""
import os, sys, time
import torch
a = torch.randn(1600, 80640, device="cuda", dtype=torch.bfloat16)
b = torch.randn(80640, 6400, device="cuda", dtype=torch.bfloat16)
def tm():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): torch.mm(a, b, out_dtype=torch.float32)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10 * 1e6
torch.mm(a, b, out_dtype=torch.float32)
print("before env: %.1f us" % tm(), flush=True)
os.environ["TENSILE_STREAMK_DATA_PARALLEL"] = "1"
print("after env:  %.1f us" % tm(), flush=True)
