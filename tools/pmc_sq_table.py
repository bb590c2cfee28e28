"""Per-kernel SQ table from two rocprofv3 --pmc passes (the counter sets of tools/gpu/r3_final.sh):
  mfma%    = SQ_VALU_MFMA_BUSY_CYCLES per SIMD (1024 SIMDs) / kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs)
  active / wait / stall = share of the waves' cycles (SQ_WAVE_CYCLES) spent issuing (SQ_ACTIVE_INST_ANY), parked in s_waitcnt or
             s_barrier (SQ_WAIT_ANY), and the rest (issue-stalled)
  valu/mfma = non-MFMA VALU instructions per MFMA;  ldsconf% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
Usage: python tools/pmc_sq_table.py <pass1 dir> <pass2 dir>"""
import collections
import csv
import glob
import re
import sys


def collect(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"^void aldm::", "", r["Kernel_Name"])
            n = re.sub(r"\(.*$", "", n)
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {n: {c: sum(v) / len(v) for c, v in cs.items()} | {"_n": max(len(v) for v in cs.values())} for n, cs in acc.items()}


def main():
    a, b = collect(sys.argv[1]), collect(sys.argv[2])
    rows = []
    for n in a:
        if n not in b:
            continue
        p, q = a[n], b[n]
        cyc = q.get("GRBM_GUI_ACTIVE", 0) / 8
        wc = p.get("SQ_WAVE_CYCLES", 0)
        if not cyc or not wc:
            continue
        mfma = p.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc * 100
        act = p.get("SQ_ACTIVE_INST_ANY", 0) / wc * 100
        wait = p.get("SQ_WAIT_ANY", 0) / wc * 100
        im = q.get("SQ_INSTS_MFMA", 0)
        vpm = f"{(p.get('SQ_INSTS_VALU', 0) - im) / im:9.1f}" if im else "        -"
        lds = q.get("SQ_LDS_IDX_ACTIVE", 0)
        conf = q.get("SQ_LDS_BANK_CONFLICT", 0) / lds * 100 if lds else 0.0
        rows.append((cyc * p["_n"], n, p["_n"], mfma, act, wait, 100 - act - wait, vpm, conf))
    rows.sort(reverse=True)
    print(f"{'kernel':58s} {'n':>5s} {'mfma%':>6s} {'active%':>7s} {'wait%':>6s} {'stall%':>6s} {'valu/mfma':>9s} {'ldsconf%':>8s}")
    for _, n, cnt, mfma, act, wait, stall, vpm, conf in rows:
        print(f"{n[:58]:58s} {cnt:5d} {mfma:6.1f} {act:7.1f} {wait:6.1f} {stall:6.1f} {vpm} {conf:8.1f}")


if __name__ == "__main__":
    main()
