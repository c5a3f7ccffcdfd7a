#!/usr/bin/env python3
"""tools/kernel_timeline.py trace_dir [n] [kernel name part] -- the last kernels of a rocprofv3 --kernel-trace run in launch
order with their start offsets and durations (ns): shows the gaps between the launches of one step."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 3:       # window that ends a few launches after the last kernel whose name contains argv[3]
    hit = max(i for i, r in enumerate(rows) if sys.argv[3] in r["Kernel_Name"])
    rows = rows[max(0, hit + 4 - n):hit + 4]
else:
    rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(r["Kernel_Name"][:60].ljust(60), int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
