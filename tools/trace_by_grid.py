"""Aggregate a rocprofv3 kernel_trace.csv by (kernel, grid, workgroup): calls, total ms, avg us — the
in-graph per-shape durations (eager event timing of small kernels is host-bound).
Usage: python tools/trace_by_grid.py <dir-with-kernel_trace.csv> [top_n]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
agg = defaultdict(lambda: [0, 0])
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("aldm::", "")
            key = (name[:44], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
            a = agg[key]
            a[0] += 1
            a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print(f"{'kernel':44s} {'grid(x,y,z threads)':>22s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>8s} {'pct':>5s}")
for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{key[0]:44s} {','.join(key[1:]):>22s} {n:7d} {t/1e6:9.3f} {t/n/1e3:8.2f} {100*t/tot:5.1f}")
print(f"TOTAL {tot/1e6:.2f} ms over {sum(v[0] for v in agg.values())} dispatches")
