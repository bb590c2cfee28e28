"""attribute the `__amd_rocclr_copyBuffer` dispatches of a rocprofv3 kernel trace of
`bench.py --warmup 1 --steps 1 --ddim-steps S`.  Everything up to the last weight-packing kernel (pack_weight / pack_split)
is the one-time set-up of the warm-up job (weight uploads = blit copies, packing); what comes after is steady state: the
timed job, whose DDIM steps are graph replays.  Usage: python tools/trace_copy_attrib.py <dir with *kernel_trace.csv> S"""
import csv
import glob
import sys

d, S = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
last_pack = max((i for i, r in enumerate(rows) if "pack_weight_kernel" in r[2] or "pack_split_kernel" in r[2]), default=-1)
setup, steady = rows[: last_pack + 1], rows[last_pack + 1:]
cp = lambda rs: [r for r in rs if "copyBuffer" in r[2]]
print(f"{len(rows)} dispatches; set-up part (up to the last weight-packing kernel): {len(setup)} dispatches, "
      f"{len(cp(setup))} copyBuffer ({sum(e - s for s, e, _ in cp(setup)) / 1e6:.2f} ms)")
st = cp(steady)
print(f"steady state (after it: the rest of the warm-up job + the timed job, {S} DDIM steps each): {len(steady)} dispatches, "
      f"{len(st)} copyBuffer ({sum(e - s for s, e, _ in st) / 1e6:.3f} ms)")
# inside the replayed steps: the dispatches between two consecutive ddim_step_indexed kernels
idx = [i for i, r in enumerate(steady) if "ddim_step_indexed" in r[2]]
if len(idx) >= 3:
    per = []
    for a, b in zip(idx[:-1], idx[1:]):
        seg = steady[a + 1:b + 1]
        if len(seg) > 200:   # a DDIM step (a job boundary in between would contain VAE / vocoder kernels too: skip those)
            if not any("igemm_kernel<128, 128, 2, 4, 3" in r[2] for r in seg):
                per.append((len(seg), len(cp(seg))))
    if per:
        print(f"{len(per)} replayed DDIM steps found: {sum(p[0] for p in per) / len(per):.0f} dispatches per step, "
              f"copyBuffer per step: min {min(p[1] for p in per)} max {max(p[1] for p in per)}")
