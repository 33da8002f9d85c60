#!/usr/bin/env python3
"""A/B of the greedy leg of bench.py (MM_DECODE_SPLIT=0|1 etc. from the environment)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
r = bench.greedy_bench(dev, 1, 0)
print("tok/s", r["tokens_per_s_per_gpu"], "total s", r["seconds"], "prefill s", r["prefill_seconds"], "ms/step", r["decode_loop_roofline"]["ms_per_decode_step"],
      "loop frac", r["decode_loop_roofline"]["frac"], "fp32 tok/s", r["fp32"]["tokens_per_s_per_gpu"], flush=True)
k4 = bench.decode_roofline(dev)
print("K4", k4["launch_us"], k4["achieved"], k4["frac"], flush=True)
