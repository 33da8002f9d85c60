#!/usr/bin/env python3
"""Lab only (library built with -DRES_LAB, MM_ATTN_RES_DBG=5): per-wave s_memrealtime stamps (100 MHz) of the resident forward
attention kernel -- kernel entry, after the load barrier, after each 16-row group, exit -- to see where a workgroup's time goes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_amd import ops  # noqa: E402

B, H, d, S, P = 32, 16, 100, 315, 221
dev = torch.device("cuda:0")
E = H * d
torch.manual_seed(0)
qkv = (torch.rand(B, S, 3, E, device=dev) * 2 - 1).to(torch.bfloat16)
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
stamps = torch.zeros(B * H, 16, 8, dtype=torch.int64, device=dev)
fake = stamps.view(torch.float32)
sin_t = torch.zeros(8, device=dev)
for _ in range(3):
    stamps.zero_()
    ops.prefix_attn_fwd(q, k, v, H, 0, P, fake, sin_t, True, save_lse=True)
torch.cuda.synchronize()
st = stamps.cpu().double()
t0 = st[:, :, 0][st[:, :, 0] > 0].min()
us = lambda x: (x - t0) / 100.0
start, bar = us(st[:, :, 0]), us(st[:, :, 1])
last = st.max(dim=2).values
end = us(last)
print(f"workgroups {B * H}; kernel span {float(end.max()):.1f} us")
print(f"wave start: min {float(start.min()):.1f} median {float(start.median()):.1f} max {float(start.max()):.1f}")
print(f"load phase (entry -> barrier) per wave: median {float((bar - start).median()):.2f} us, max {float((bar - start).max()):.2f}")
wg_start, wg_end = start.min(dim=1).values, end.max(dim=1).values
print(f"workgroup lifetime: median {float((wg_end - wg_start).median()):.2f} us, max {float((wg_end - wg_start).max()):.2f}")
first = wg_start < 2.0
print(f"first-round workgroups {int(first.sum())}: lifetime median {float((wg_end - wg_start)[first].median()):.2f}; later: start median "
      f"{float(wg_start[~first].median()):.2f}, lifetime median {float((wg_end - wg_start)[~first].median()):.2f}")
# per group durations: stamps 2.. are group ends
g1 = us(st[:, :, 2]) - bar
print(f"first group of a wave (barrier -> group end): median {float(g1[st[:, :, 2] > 0].median()):.2f} us, max {float(g1[st[:, :, 2] > 0].max()):.2f}")
idle = wg_end[:, None] - end
print(f"wave idle before its workgroup ends: median {float(idle.median()):.2f} us, mean {float(idle.mean()):.2f}")
