"""Ring depth of the loader-wave 64x64 tile on the 1024-row (64-token level) GEMMs, 3-part images: per-launch time, HIP-graph timed,
for LDS rings of 3 (the tuned tables' choice), 4 and 6 k-tiles.  Usage: python tools/lw_ring_probe.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402

ops.set_mma("bf16x6")


def timed(fn, reps=40):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


for M, K, N, res in [(1024, 640, 640, True), (1024, 640, 640, False), (1024, 1280, 640, False), (1024, 384, 640, False),
                     (4096, 384, 384, True)]:
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(1, M, K, generator=gen)
    w = torch.randn(N, K, generator=gen) / math.sqrt(K)
    b = torch.randn(N, generator=gen)
    r = torch.randn(1, M, N, generator=gen).cuda() if res else None
    xs, pw = ops.split_rows(x.cuda()), ops.pack_conv(w, b)
    ref = x.double() @ w.double().t() + b.double() + (r.double().cpu() if res else 0)
    line = f"M{M} K{K} N{N}{' +res' if res else ''}: auto {timed(lambda: ops.linear(xs, pw, res=r)):6.2f} us"
    for st in (203, 204, 206):
        ops.igemm_force(64, 64, 1, 0, st)
        try:
            y = ops.linear(xs, pw, res=r)
            err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
            line += f" | lw{st - 200} {timed(lambda: ops.linear(xs, pw, res=r)):6.2f} us (err {err:.1e})"
        finally:
            ops.igemm_force(0, 0, 0)
    print(line, flush=True)
