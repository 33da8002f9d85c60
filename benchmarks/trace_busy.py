#!/usr/bin/env python3
"""trace_busy.py <kernel_trace.csv> [t0_frac t1_frac]: GPU busy time (union of kernel intervals) against the wall span of a window
of the trace, plus the kernels of that window grouped by name: tells a host-bound loop (large idle share) from a GPU-bound one."""
import csv
import sys
from collections import defaultdict

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
lo, hi = rows[0][0], rows[-1][1]
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1.0)
w0, w1 = lo + (hi - lo) * f0, lo + (hi - lo) * f1
sel = [r for r in rows if r[0] >= w0 and r[1] <= w1]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += (cur_e - cur_s) if cur_e is not None else 0
span = sel[-1][1] - sel[0][0]
print(f"window {f0:.2f}-{f1:.2f}: {len(sel)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms ({busy / span:.1%}), "
      f"sum of durations {sum(e - s for s, e, _ in sel) / 1e6:.3f} ms")
g = defaultdict(list)
for s, e, n in sel:
    g[n[:90]].append(e - s)
for n, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"  {sum(v) / 1e6:8.3f} ms  n={len(v):5d}  avg {sum(v) / len(v) / 1e3:7.1f} us  {n}")
