#!/usr/bin/env python3
"""Average duration per (kernel, grid) from a rocprofv3 --kernel-trace CSV: separates the shapes a kernel runs on (the stats CSV averages them).
    python3 scripts/trace_by_shape.py trace.csv [grep]"""
import csv, gzip, sys, collections
f = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
op = gzip.open if f.endswith(".gz") else open
acc = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(op(f, "rt")):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if pat and pat not in name:
        continue
    key = (name[:60], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r.get("Grid_Size_Y", 1)) // max(1, int(r.get("Workgroup_Size_Y", 1))))
    a = acc[key]
    a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
for (name, gx, gy), (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{t / n / 1e3:9.1f} us  x{n:5d}  total {t / 1e6:8.2f} ms   grid {gx} x {gy}   {name}")
