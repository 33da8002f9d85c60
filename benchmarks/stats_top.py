#!/usr/bin/env python3
"""stats_top.py <dir or kernel_stats.csv> [n]: the n largest rows of a rocprofv3 --stats kernel summary (total ms, calls, mean us, name)."""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms in {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:n]:
    print(f"{float(r['TotalDurationNs']) / 1e6:8.2f} ms {r['Calls']:>6s} x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:120]}")
