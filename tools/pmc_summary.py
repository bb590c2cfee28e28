"""Average the counters of a rocprofv3 --pmc run per kernel: python tools/pmc_summary.py <dir> [name regex]."""
import collections
import csv
import glob
import re
import sys

pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if pat and not pat.search(n):
            continue
        acc[n[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print(n)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
