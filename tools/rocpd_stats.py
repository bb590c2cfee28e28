"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table.
Usage: python tools/rocpd_stats.py <results.db> [top_n]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        name = re.sub(r"\(.*$", "", name)
        name = name.replace("void ", "")
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += (e - s)
    total = sum(v[1] for v in agg.values())
    print(f"{'kernel':70s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>9s} {'pct':>6s}")
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name[:70]:70s} {n:8d} {t/1e6:10.3f} {t/n/1e3:9.2f} {100*t/total:6.2f}")
    print(f"{'TOTAL':70s} {sum(v[0] for v in agg.values()):8d} {total/1e6:10.3f}")


if __name__ == "__main__":
    main()
