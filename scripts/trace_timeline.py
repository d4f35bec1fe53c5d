#!/usr/bin/env python3
"""Print a per-kernel timeline (start offset, duration, queue, grid, VGPRs) from a rocprofv3 --kernel-trace CSV (.csv or .csv.gz).

    python scripts/trace_timeline.py trace.csv.gz [--from K] [--count N] [--grep name]
Used to see which launches of two HIP streams actually overlap."""
import csv, gzip, sys, argparse
ap = argparse.ArgumentParser()
ap.add_argument("file"); ap.add_argument("--skip", type=int, default=0); ap.add_argument("--count", type=int, default=200)
ap.add_argument("--after", default=None, help="start at the LAST dispatch whose name contains this")
a = ap.parse_args()
op = gzip.open if a.file.endswith(".gz") else open
rows = list(csv.DictReader(op(a.file, "rt")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i0 = a.skip
if a.after:
    i0 = max(i for i, r in enumerate(rows) if a.after in r["Kernel_Name"])
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = {}
for r in rows[i0:i0 + a.count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:44]
    q = r["Queue_Id"]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{q} grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):6d} v{r['VGPR_Count']:>3}+{r['Accum_VGPR_Count']:<3} lds {int(r['LDS_Block_Size']) // 1024:3d}K  {name}")
