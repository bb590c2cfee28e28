"""GroupNorm launch times on the UNet's shapes (16 samples = batch 8 with CFG), 20 launches per HIP-graph replay.
Default: the statistics alone (ops.gn_stats); --split: statistics + apply + SiLU + operand split (ops.gn_split, what the UNet calls).
Environment switches (read once per process by norm.hip): ALDM_GN_FUSED_MAX=<elements> moves the boundary between the one-launch
(group-sliced) and the chunked two-launch form; ALDM_GN_SPLIT_FUSED=0 keeps the split in its own launch.  (ALDM_GN_THREADS /
ALDM_GN_MIN_BLOCKS of profiles/r03_gn_split_bench.txt belong to an experiment that was not kept: its diff is in profiles/.)"""
import os
import sys

import torch

sys.path.insert(0, ".")
from audioldm2_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
B = 16
tot = 0.0
for H, W, C1, C2, n in ((32, 2, 640, 0, 6), (32, 2, 640, 640, 3), (32, 2, 1280, 0, 4), (32, 2, 640, 384, 1),
                        (64, 4, 384, 0, 8), (64, 4, 384, 384, 2), (64, 4, 768, 0, 3), (64, 4, 640, 384, 1), (64, 4, 384, 256, 1),
                        (128, 8, 256, 0, 8), (128, 8, 256, 256, 2), (128, 8, 512, 0, 3), (128, 8, 384, 256, 1), (128, 8, 256, 128, 1),
                        (256, 16, 128, 0, 6), (256, 16, 128, 128, 3), (256, 16, 256, 0, 3), (256, 16, 256, 128, 1)):
    x = torch.randn(B, H, W, C1, generator=g).cuda()
    x2 = torch.randn(B, H, W, C2, generator=g).cuda() if C2 else None
    ga, be = torch.ones(C1 + C2).cuda(), torch.zeros(C1 + C2).cuda()
    if "--split" in sys.argv:
        fn = lambda: ops.gn_split(x, ga, be, groups=32, eps=1e-5, x2=x2, act=ops.ACT_SILU)
    else:
        fn = lambda: ops.gn_stats(x, ga, be, groups=32, eps=1e-5, x2=x2)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(20):
                keep = fn()
    torch.cuda.synchronize()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3 / 100
    tot += t * n
    print(f"{'gn_split' if '--split' in sys.argv else 'gn_stats'} P={H * W:5d} C={C1}+{C2}: {t:6.1f} us  (~{n} per UNet pass)", flush=True)
print(f"weighted total {tot:.0f} us per UNet pass   [" + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("ALDM_GN")) + "]")
