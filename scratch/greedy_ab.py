import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
d = torch.device("cuda", 0)
r = bench.greedy_bench(d, 1, 0); print("B=1024", r["tokens_per_s_per_gpu"], r["seconds"], flush=True)
r = bench.greedy_bench(d, 8, 0); print("B=128", r["tokens_per_s_per_gpu"], r["seconds"], flush=True)
