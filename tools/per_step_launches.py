#!/usr/bin/env python3
"""Per-step kernel launches of a bench command: two rocprofv3 kernel traces with different step counts, differenced (everything that
does not scale with the steps -- model build, packing plans, warm-up -- cancels).
    python tools/per_step_launches.py <trace_small.csv> <steps_small> <trace_large.csv> <steps_large>"""
import csv, sys
from collections import defaultdict


def load(path):
    n, t = defaultdict(int), defaultdict(float)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        n[k] += 1
        t[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return n, t


(n0, t0), s0 = load(sys.argv[1]), int(sys.argv[2])
(n1, t1), s1 = load(sys.argv[3]), int(sys.argv[4])
ds = s1 - s0
rows = []
for k in n1:
    dn, dt = n1[k] - n0.get(k, 0), t1[k] - t0.get(k, 0.0)
    if dn > 0:
        rows.append((dt / ds, dn / ds, k))
rows.sort(reverse=True)
tot_n = sum(r[1] for r in rows); tot_t = sum(r[0] for r in rows)
torch_n = sum(r[1] for r in rows if "at::native" in r[2] or "rocclr" in r[2] or "rocprim" in r[2])
torch_t = sum(r[0] for r in rows if "at::native" in r[2] or "rocclr" in r[2] or "rocprim" in r[2])
print(f"# per step: {tot_n:.1f} launches, {tot_t / 1e3:.3f} ms of kernel time; torch / runtime glue (at::native, rocclr, rocprim): {torch_n:.1f} launches, {torch_t:.1f} us")
print(f"{'us/step':>10} {'calls/step':>10}  kernel")
for dt, dn, k in rows:
    print(f"{dt:10.1f} {dn:10.2f}  {k[:150]}")
