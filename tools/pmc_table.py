"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel+grid: mean counter values.
Usage: python tools/pmc_table.py <dir>"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
agg = defaultdict(lambda: defaultdict(list))
order = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "")
            key = (name, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
            if key not in agg:
                order.append(key)
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in order:
    cs = agg[key]
    n = max(len(v) for v in cs.values())
    print(f"{key[0][:60]} grid={key[1]} wg={key[2]} dispatches={n}")
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} mean {sum(v)/len(v):16.1f}")
