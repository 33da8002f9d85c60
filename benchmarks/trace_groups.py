#!/usr/bin/env python3
"""trace_groups.py <kernel_trace.csv> <out.csv>: rocprofv3 kernel-trace rows grouped by (kernel, grid, workgroup), so that a
kernel launched at several shapes in one command (bench.py: config-batch launches and the roofline-batch launches of the
same kernel) gets one average per shape -- the per-shape averages are what bench.py's live HIP-event timings must match."""
import csv
import sys
from collections import defaultdict

g = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    g[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(g.items(), key=lambda kv: -sum(kv[1]))
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Kernel_Name", "Grid_X", "Grid_Y", "Grid_Z", "Workgroup_X", "Calls", "TotalNs", "AverageNs", "MinNs", "MaxNs"])
    for (name, gx, gy, gz, wx), v in rows:
        w.writerow([name[:160], gx, gy, gz, wx, len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v)])
