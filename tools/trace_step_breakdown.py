"""Where one replayed DDIM step's time goes, from a rocprofv3 kernel trace of `bench.py --warmup 1 --steps 1 --ddim-steps S`:
the dispatches between two consecutive `ddim_step_indexed` kernels are one UNet pass + its DDIM update.  Averages over the
replayed steps of the timed job: kernel time by category and by instantiation, dispatches, and the time no kernel runs (gaps).
Usage: python tools/trace_step_breakdown.py <dir with *kernel_trace.csv> [top-N instantiations]"""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if "ddim_step_indexed" in r[2]]
segs = [rows[a + 1:b + 1] for a, b in zip(idx[:-1], idx[1:])]
# a replayed step: no weight packing (the first eager pass), and the dispatch count every replay has (a job boundary also holds the
# VAE / vocoder kernels; the drawer thread's noise upload may add one or two copyBuffer dispatches)
segs = [s for s in segs if len(s) > 200 and not any("pack_" in r[2] for r in s)]
if not segs:
    sys.exit("no replayed DDIM step found")
med = sorted(len(s) for s in segs)[len(segs) // 2]
steps = [s for s in segs if abs(len(s) - med) <= 3]
steps = steps[len(steps) // 2:]   # the timed job's


def short(name):
    name = name.replace("void aldm::", "").replace("aldm::", "")
    return re.sub(r"\(.*$", "", name)


def category(n):
    if n.startswith("igemm_dma_halo"):
        return "GEMM: halo-patch 3x3 conv"
    if n.startswith("igemm_dma_os"):
        return "GEMM: operand-stationary (K <= 384)"
    if n.startswith("igemm_dma_lw"):
        return "GEMM: loader-wave"
    if n.startswith("igemm_dma"):
        return "GEMM: classic DMA-fed"
    if n.startswith("igemm_reduce"):
        return "GEMM: split-K reduce"
    if n.startswith("igemm"):
        return "GEMM: register-staged"
    if n.startswith("attention"):
        return "attention"
    if n.startswith("gn_"):
        return "GroupNorm (+SiLU, split)"
    if n.startswith("layernorm"):
        return "LayerNorm (+split)"
    if n.startswith("split_rows"):
        return "split_rows"
    return "other (embeddings, DDIM update, elementwise)"


ns = len(steps)
by_k, by_c = defaultdict(lambda: [0, 0]), defaultdict(lambda: [0, 0])
span = busy = 0
for seg in steps:
    span += seg[-1][1] - seg[0][0]
    for s, e, n in seg:
        k = short(n)
        by_k[k][0] += 1; by_k[k][1] += e - s
        c = category(k)
        by_c[c][0] += 1; by_c[c][1] += e - s
        busy += e - s
print(f"{ns} replayed DDIM steps: {sum(len(s) for s in steps) / ns:.0f} dispatches per step, first kernel start -> last kernel end "
      f"{span / ns / 1e6:.3f} ms, kernel time {busy / ns / 1e6:.3f} ms, no kernel running {max(span - busy, 0) / ns / 1e6:.3f} ms")
print(f"\n{'category':48s} {'launches':>9s} {'ms/step':>9s} {'share':>7s} {'avg us':>8s}")
for c, (n, t) in sorted(by_c.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:48s} {n / ns:9.1f} {t / ns / 1e6:9.3f} {100.0 * t / busy:6.1f}% {t / n / 1e3:8.2f}")
print(f"\n{'instantiation':64s} {'launches':>9s} {'ms/step':>9s} {'share':>7s} {'avg us':>8s}")
for k, (n, t) in sorted(by_k.items(), key=lambda kv: -kv[1][1])[:TOP]:
    print(f"{k[:64]:64s} {n / ns:9.1f} {t / ns / 1e6:9.3f} {100.0 * t / busy:6.1f}% {t / n / 1e3:8.2f}")
