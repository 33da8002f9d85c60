#!/usr/bin/env python3
"""trace_step.py <kernel_trace.csv>: timeline of the LAST full train step in a rocprofv3 kernel trace of bench.py --no-extras:
per queue busy time, union busy time, idle gaps, and the phase boundaries (forward / backward / optimizer) found from kernel names."""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
rows.sort()
# steps are delimited by the optimizer's og_adamw_k launch at the END of a step: with the decoder updates deferred into the next
# forward there are more launches per step (the deferred ones follow within a few ms), so only a launch whose predecessor is
# more than 8 ms older counts
adam = [i for i, r in enumerate(rows) if r[2].startswith("og_adamw_k")]
ends = [i for n, i in enumerate(adam) if n == 0 or rows[i][0] - rows[adam[n - 1]][0] > 8_000_000]
assert len(ends) >= 3, "need at least 3 optimizer launches in the trace"
if rows[-1][0] - rows[ends[-1]][0] < 8_000_000 and len(adam) and adam[-1] != ends[-1]:
    ends = ends[:-1] if len(ends) > 3 else ends       # the last step's deferred launches came from the final flush: use the one before
i0, i1 = ends[-2] + 1, ends[-1] + 1
step = rows[i0:i1]
t0, t1 = step[0][0], step[-1][1]
print(f"step: {len(step)} kernels, span {(t1 - t0) / 1e6:.3f} ms (prev adamw end -> this adamw end: {(step[-1][1] - rows[ends[-2]][1]) / 1e6:.3f} ms)")


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if ce is None or s > ce:
            if ce is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ((ce - cs) if ce is not None else 0)


byq = defaultdict(list)
for s, e, n, q, st in step:
    byq[(q, st)].append((s, e))
print(f"union busy {union([(s, e) for s, e, *_ in step]) / 1e6:.3f} ms; sum of durations {sum(e - s for s, e, *_ in step) / 1e6:.3f} ms")
for q, iv in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    print(f"  queue/stream {q}: {len(iv)} kernels, busy {union(iv) / 1e6:.3f} ms, first +{(min(s for s, _ in iv) - t0) / 1e6:.3f} ms, last end +{(max(e for _, e in iv) - t0) / 1e6:.3f} ms")
# phase marks
def first(pred):
    return next((r for r in step if pred(r[2])), None)
marks = [("first bwd attention (dkv/dq)", first(lambda n: "bwd_dq" in n or "bwd_dkv" in n)),
         ("softmax/CE backward (nll)", first(lambda n: "nll_loss_backward" in n or "log_softmax_backward" in n)),
         ("stem backward", first(lambda n: "stem_ln_bwd" in n)),
         ("og_reduce", first(lambda n: n.startswith("og_reduce_k")))]
for nm, r in marks:
    if r:
        print(f"  {nm}: +{(r[0] - t0) / 1e6:.3f} ms")
# idle gaps of the union
iv = sorted((s, e) for s, e, *_ in step)
gaps, ce = [], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        gaps.append((s - ce, ce - t0))
    ce = max(ce, e)
gaps.sort(reverse=True)
print(f"idle: total {sum(g for g, _ in gaps) / 1e6:.3f} ms in {len(gaps)} gaps; largest: " + ", ".join(f"{g / 1e3:.0f}us@+{a / 1e6:.2f}ms" for g, a in gaps[:8]))
# kernels between the stem's backward and the optimizer's first launch (the end-of-backward tail)
stem = first(lambda n: "stem_ln_bwd" in n)
ogr = first(lambda n: n.startswith("og_reduce_k"))
if stem and ogr:
    print("tail kernels (stem backward -> og_reduce):")
    for s, e, n, q, st in step:
        if s >= stem[0] and s <= ogr[0]:
            print(f"   +{(s - t0) / 1e6:7.3f} ms  {(e - s) / 1e3:7.1f} us  q{q}  {n[:100]}")
# forward phase (step start -> first loss-backward kernel): kernel time by name, and the gaps inside it
ce = first(lambda n: "nll_loss_backward" in n or "log_softmax_backward" in n)
if ce:
    fw = [r for r in step if r[0] < ce[0]]
    print(f"forward: {len(fw)} kernels, span {(fw[-1][1] - fw[0][0]) / 1e6:.3f} ms, busy {union([(s, e) for s, e, *_ in fw]) / 1e6:.3f} ms")
    g = defaultdict(list)
    for s, e, n, q, st in fw:
        g[n[:70]].append(e - s)
    for n, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:22]:
        print(f"   {sum(v) / 1e6:7.3f} ms  n={len(v):4d}  avg {sum(v) / len(v) / 1e3:7.1f} us  {n}")
    bw = [r for r in step if r[0] >= ce[0] and not r[2].startswith("og_")]
    print(f"backward (both queues): {len(bw)} kernels, busy {union([(s, e) for s, e, *_ in bw]) / 1e6:.3f} ms, sum {sum(e - s for s, e, *_ in bw) / 1e6:.3f} ms")
    g = defaultdict(list)
    for s, e, n, q, st in bw:
        g[(q, n[:70])].append(e - s)
    for (q, n), v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:40]:
        print(f"   q{q} {sum(v) / 1e6:7.3f} ms  n={len(v):4d}  avg {sum(v) / len(v) / 1e3:7.1f} us  {n}")
