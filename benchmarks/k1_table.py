#!/usr/bin/env python3
"""bench.py's K1 / K1b all-shape table alone (the roofline_k1_all_shapes leg), in the layout of profiles/r03_k1_all_shapes.txt."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for e in bench.k1_all_shapes(torch.device("cuda:0")):
    for d in ("fwd", "bwd"):
        print(f"K1 {d} {e['dtype']:8s} B{e['B']} C{e['C']} L{e['L']}: {e[d + '_us']:8.1f} us {e[d + '_GBps']:7d} GB/s ({e[d + '_frac']:.3f} of 8 TB/s)", flush=True)
