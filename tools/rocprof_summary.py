#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace output (rocpd sqlite .db or *_kernel_trace.csv) into the per-kernel
summary table committed under profiles/ (calls, total/avg/min/max duration, share)."""
import csv
import sqlite3
import sys


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    return list(cur.execute("select name, end - start from kernels"))


def rows_from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    path = sys.argv[1]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    agg = {}
    for name, d in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"# source: {path}\n# total kernel time {tot / 1e6:.3f} ms over {len(rows)} dispatches")
    print(f"{'kernel':100s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[:100]:100s} {a[0]:6d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:10.1f} {a[2] / 1e3:9.1f} {a[3] / 1e3:9.1f} {100 * a[1] / tot:6.2f}")


if __name__ == "__main__":
    main()
