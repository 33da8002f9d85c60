#!/usr/bin/env python3
"""The train step with every bucket all-reduce forced through RCCL on ONE rank (bench.py's rccl_world1 leg alone), for
kernel traces: where the 3 ms between this leg and the plain step go."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

c = bench.CONFIGS["local_config"]
device = torch.device("cuda:0")
torch.cuda.set_device(device)
print(bench.rccl_world1_leg(c, device, steps=int(os.environ.get("STEPS", "8"))))
