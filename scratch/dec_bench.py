import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
d = torch.device("cuda", 0)
for _ in range(2):
    r = bench.decode_roofline(d)
    print(r["launch_us"], r["achieved"], flush=True)
r = bench.decode_roofline(d, B=128)
print("B=128", r["launch_us"], r["achieved"], flush=True)
