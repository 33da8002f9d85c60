"""Timeline analysis of a rocprofv3 kernel trace of bench.py: per step, wall time, idle time (no kernel on any
queue), time with one / two queues busy, per-queue busy time, and the phases (forward / backward / optimizer)."""
import csv
import sys

f = sys.argv[1]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
# steps are delimited by og_adamw_k (last kernel of a step)
ends = [i for i, r in enumerate(rows) if r[3].startswith("og_adamw_k")]
print("steps found", len(ends))
for si in range(max(1, len(ends) - 3), len(ends)):
    a, b = ends[si - 1] + 1, ends[si] + 1
    ks = rows[a:b]
    t0, t1 = rows[ends[si - 1]][1], ks[-1][1]
    # sweep
    ev = []
    for s, e, q, n in ks:
        ev.append((max(s, t0), 1, q)); ev.append((e, -1, q))
    ev.sort()
    depth = 0; last = t0; busy = {0: 0, 1: 0, 2: 0}
    for t, d, q in ev:
        busy[min(depth, 2)] += t - last
        last = t; depth += d
    qbusy = {}
    for s, e, q, n in ks:
        qbusy[q] = qbusy.get(q, 0) + e - s
    # phase boundaries: first backward kernel = first kernel after the loss (nll_loss_backward)
    def first(pred):
        for s, e, q, n in ks:
            if pred(n):
                return s
        return None
    tb = first(lambda n: "nll_loss_backward" in n or "log_softmax_backward" in n)
    to = first(lambda n: n.startswith("og_reduce_k"))
    print(f"step {si}: wall {(t1 - t0) / 1e6:.3f} ms  idle {busy[0] / 1e6:.3f}  one-queue {busy[1] / 1e6:.3f}  two+ {busy[2] / 1e6:.3f}"
          f"  | fwd {(tb - t0) / 1e6:.3f}  bwd {(to - tb) / 1e6:.3f}  opt {(t1 - to) / 1e6:.3f}  | queue busy "
          + " ".join(f"q{q}:{v / 1e6:.2f}" for q, v in sorted(qbusy.items())))
    # idle gaps per phase
    def idle_in(lo, hi):
        ev2 = sorted((max(s, lo), min(e, hi)) for s, e, q, n in ks if e > lo and s < hi)
        cur = lo; idle = 0
        for s, e in ev2:
            if s > cur:
                idle += s - cur
            cur = max(cur, e)
        return idle + max(0, hi - cur)
    print(f"   idle: fwd {idle_in(t0, tb) / 1e6:.3f}  bwd {idle_in(tb, to) / 1e6:.3f}  opt {idle_in(to, t1) / 1e6:.3f}")
    if si == len(ends) - 1 and len(sys.argv) > 2:
        # per-queue dump of the backward phase: main-queue kernels and whether the other queue was busy
        for s, e, q, n in ks:
            if s >= tb - 1000 and s < to:
                print(f"   {(s - tb) / 1e3:9.1f} us +{(e - s) / 1e3:7.1f} q{q} {n[:90]}")
