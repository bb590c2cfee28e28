"""Do two independent kernel chains captured as PARALLEL branches of one HIP graph overlap on this stack?  Two chains of N
DMA-fed GEMM launches (each launch depends on the previous one of its chain through its output buffer ordering on the
stream): one chain alone, the two chains one after the other on one stream, the two chains as two branches (two streams forked
from / joined to the capture stream).  If branches overlap and the kernels are latency bound, `branches` ~ `one chain`;
if the kernels already fill the machine, `branches` ~ `sequential`.  Usage: python tools/branch_concurrency.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402

ops.set_mma("bf16x3")


def chain(M, K, N, n, seed):
    g = torch.Generator().manual_seed(seed)
    x = ops.split_rows(torch.randn(1, M, K, generator=g).cuda())
    pw = ops.pack_conv(torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g))
    out = torch.empty((1, M, N), device="cuda")
    ops.linear(x, pw, out=out.view(1, 1, M, N))   # warm: builds the weight image

    def run():
        for _ in range(n):
            ops.linear(x, pw, out=out.view(1, 1, M, N))
    return run


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, (M, K, N) in {"level-3 proj 1024x640->640 (160 blocks)": (1024, 640, 640),
                        "level-2 proj 4096x384->384 (384 blocks)": (4096, 384, 384),
                        "level-1 proj 16384x256->256 (512 blocks)": (16384, 256, 256),
                        "level-1 geglu-size 16384x256->2048": (16384, 256, 2048)}.items():
    n = 40
    a, b = chain(M, K, N, n, 1), chain(M, K, N, n, 2)

    def seq():
        a()
        b()

    def par():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            a()
        with torch.cuda.stream(s2):
            b()
        cur.wait_stream(s1)
        cur.wait_stream(s2)
    t1, ts, tp = timed(a), timed(seq), timed(par)
    print(f"{name}: one chain of {n}: {t1:7.1f} us ({t1 / n:.1f}/launch)   two chains sequential: {ts:7.1f} us   as two graph "
          f"branches: {tp:7.1f} us  -> overlap {(ts - tp) / max(ts - t1, 1e-9) * 100:.0f} % of the second chain hidden", flush=True)
