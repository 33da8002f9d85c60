#!/usr/bin/env python3
"""decode_trace.py <kernel_trace.csv>: the decode loop of one generate() call seen through rocprofv3's kernel trace.
Takes the LAST run of decode-attention launches, and reports per kernel name (count, mean us), the sum of kernel times,
the union of busy intervals and the time two kernels really ran beside each other, per decode step."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows))
k4 = [i for i, e in enumerate(ev) if "decode_attn" in e[2]]
if not k4:
    sys.exit("no decode_attn launches in the trace")
# the last burst of K4 launches: walk back while the gap between K4 launches stays below 2 ms
j = len(k4) - 1
while j > 0 and ev[k4[j]][0] - ev[k4[j - 1]][0] < 2_000_000:
    j -= 1
lo, hi = k4[j], k4[-1]
seg = ev[lo:hi + 1]
t0, t1 = seg[0][0], max(e[1] for e in seg)
n_k4 = len(k4) - j
by = defaultdict(list)
for s, e, n, q in seg:
    by[n.split("(")[0][:70]].append((e - s) / 1e3)
pts = sorted([(s, 1) for s, e, n, q in seg] + [(e, -1) for s, e, n, q in seg])
busy = both = 0
depth, last = 0, pts[0][0]
for t, dlt in pts:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        both += t - last
    depth += dlt
    last = t
tot = sum(e - s for s, e, n, q in seg)
print(f"K4 launches {n_k4}, span {(t1 - t0) / 1e6:.2f} ms, kernel-time sum {tot / 1e6:.2f} ms, busy union {busy / 1e6:.2f} ms, two or more kernels at once {both / 1e6:.2f} ms")
print(f"queues: {sorted(set(q for _, _, _, q in seg))}")
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v) / 1e3:8.2f} ms {len(v):6d} x {sum(v) / len(v):8.1f} us  {n}")
