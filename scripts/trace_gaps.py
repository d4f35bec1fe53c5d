#!/usr/bin/env python3
"""Idle time between consecutive kernels of one queue in a rocprofv3 --kernel-trace CSV: for the steady-state part of a bench run
(the last N dispatches), the sum of kernel durations, the sum of the gaps between the end of a kernel and the start of the next, and
the gaps grouped by the kernel that FOLLOWS them (whose launch / ramp they are).

    python scripts/trace_gaps.py trace.csv[.gz] [--last 4000]"""
import csv, gzip, sys, argparse, collections
ap = argparse.ArgumentParser()
ap.add_argument("file"); ap.add_argument("--last", type=int, default=4000)
a = ap.parse_args()
op = gzip.open if a.file.endswith(".gz") else open
rows = list(csv.DictReader(op(a.file, "rt")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-a.last:]
busy = gap = 0
by = collections.defaultdict(lambda: [0, 0])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    if prev_end is not None:
        g = max(0, s - prev_end)
        if g < 200000:            # ignore host stalls (> 0.2 ms: bench bookkeeping between phases)
            gap += g
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:48]
            by[name][0] += g
            by[name][1] += 1
    prev_end = max(prev_end or 0, e)
print(f"{len(rows)} dispatches: kernels {busy / 1e6:.3f} ms, gaps {gap / 1e6:.3f} ms = {100 * gap / (busy + gap):.1f} % of the timeline")
for name, (g, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {g / 1e3:9.1f} us over {n:5d} launches = {g / n / 1e3:6.2f} us before each  {name}")
