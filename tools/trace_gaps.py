"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run: python tools/trace_gaps.py <dir>.
Prints, for the busiest stretch of the trace (the graph-replayed DDIM steps), the sum of kernel durations, the sum of the
gaps between one kernel's end and the next one's start, and the gap histogram."""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
rows.sort()
# keep the UNet-step kernels: everything between the first and the last attention launch
idx = [i for i, r in enumerate(rows) if "attention" in r[2]]
rows = rows[idx[0]:idx[-1] + 1]
busy = sum(e - s for s, e, _ in rows)
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
big = [g for g in gaps if g > 200_000]            # host-side pauses between steps (noise upload, python)
small = [g for g in gaps if g <= 200_000]
print(f"{len(rows)} kernels over {(rows[-1][1] - rows[0][0]) / 1e6:.2f} ms: kernel time {busy / 1e6:.2f} ms, "
      f"gaps <= 200 us: {sum(small) / 1e6:.2f} ms ({sum(small) / len(small) / 1e3:.2f} us each), "
      f"{len(big)} longer pauses: {sum(big) / 1e6:.2f} ms")
edges = [0, 500, 1000, 2000, 3000, 5000, 10000, 50000, 200000]
for lo, hi in zip(edges, edges[1:]):
    n = sum(1 for g in small if lo <= g < hi)
    print(f"  gap {lo / 1e3:5.1f}-{hi / 1e3:5.1f} us: {n}")
