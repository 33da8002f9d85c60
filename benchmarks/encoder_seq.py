#!/usr/bin/env python3
"""encoder_seq.py <kernel_trace.csv>: the kernel sequence of the LAST training pass of benchmarks/encoder_bench.py
(stem forward ... stem backward): start offset, duration, queue, grid, name -- the encoder's critical chain."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Grid_Size", r.get("Grid_Size_X", ""))) for r in rows))
bwd = [i for i, e in enumerate(ev) if "stem_ln_bwd" in e[2]]
fwd = [i for i, e in enumerate(ev) if "stem_ln_fwd" in e[2]]
if not bwd or not fwd:
    sys.exit("no stem kernels found")
hi = bwd[-1]
lo = max(i for i in fwd if i < hi)
t0 = ev[lo][0]
busy = 0
for s, e, n, q, g in ev[lo:hi + 1]:
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  q{q:>2} grid {g:>9}  {n.split('(')[0][:90]}")
print(f"span {(ev[hi][1] - t0) / 1e3:.1f} us, kernel-time sum {busy / 1e3:.1f} us, {hi - lo + 1} kernels")
