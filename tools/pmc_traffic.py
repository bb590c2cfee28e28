"""Turn two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, separate runs as MI355X_MICROARCH.md
prescribes) of the bench command into profiles/rNN_pmc_traffic.json (stamped with audioldm2_amd.lib.source_hash()): average HBM bytes per launch for
each igemm instantiation (tile x prologue mode, named as rocprofv3 names them).  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of
the bytes of wide (16 B/lane) coalesced reads (same guide) -> doubled here, raw value kept too.
Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [command the passes profiled]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def collect(root, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                m = re.search(r"igemm(?:_dma(?:_lw|_ws|_os)?)?_kernel<[^>]*>", r["Kernel_Name"])
                if not m:
                    continue
                a = acc[m.group(0)]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from audioldm2_amd.lib import source_hash, tuning_hash
    out = {"source_hash": source_hash(), "igemm_source_hash": source_hash("igemm"), "tuning_hash": tuning_hash(),
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two runs) of " +
                     (sys.argv[4] if len(sys.argv) > 4 else
                      "ALDM_NO_GRAPH=1 python bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline"),
           "units": "bytes per launch, averaged over all launches of the instantiation in that command (all prologue modes)",
           "kernels": {}}
    for k in sorted(fetch):
        n, tot = fetch[k]
        fr = tot / n * 1024.0
        wn, wt = write.get(k, [0, 0.0])
        wr = wt / wn * 1024.0 if wn else 0.0
        out["kernels"][k] = {"launches": n, "fetch_bytes_per_launch_raw": round(fr),
                             "fetch_bytes_per_launch_corrected": round(2 * fr),
                             "write_bytes_per_launch": round(wr), "hbm_bytes_per_launch": round(2 * fr + wr)}
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
