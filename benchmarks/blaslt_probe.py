#!/usr/bin/env python3
"""The hipBLASLt algorithm sweep of blaslt_probe.cpp inside a PyTorch process: the library swept is the one PyTorch bundles and
the train step calls (torch/lib/libhipblaslt.so), not /opt/rocm's."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_amd  # noqa: F401  (sets TENSILE_STREAMK_DATA_PARALLEL as the product does)
a = torch.randn(256, 256, device="cuda", dtype=torch.bfloat16)
torch.mm(a, a); torch.cuda.synchronize()          # PyTorch's hipBLASLt is loaded and initialised now
for line in open("/proc/self/maps"):
    if "hipblaslt" in line and line.rstrip().endswith(".so"):
        print("hipBLASLt in this process:", line.split()[-1]); break
del a
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libblaslt_probe.so"))
lib.run_probe.argtypes = [ctypes.c_int64]
sys.stdout.flush()
sys.exit(lib.run_probe(int(os.environ.get("TOKENS", "10080"))))
