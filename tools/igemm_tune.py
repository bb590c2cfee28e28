"""Sweep block tile x split-K for the UNet's dominant igemm shapes (effective batch 16) and print
us / TFLOP/s per variant; '*' marks what the automatic heuristic picks.
Usage (GPU box): python tools/igemm_tune.py"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402
from audioldm2_amd.ops import ACT_SILU  # noqa: E402

B = 16


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def sweep(name, fn, flops, M, N, nk):
    ops.igemm_force(0, 0, 0, 0)
    t_auto = timeit(fn)
    res = []
    for bm, bn in [(128, 128), (128, 64), (64, 128), (64, 64)]:
        if bn > 64 and N <= 64:
            continue
        for sp in [1, 2, 3, 4, 6, 8, 12, 16]:
            if sp > 1 and (nk < 8 or nk // sp < 2):
                continue
            blocks = math.ceil(M / bm) * math.ceil(N / bn) * sp
            if sp > 1 and blocks > 2048:
                continue
            for kg in ((1, 2) if (bm, bn) == (64, 64) and nk // sp >= 4 else (1,)):
                ops.igemm_force(bm, bn, sp, kg)
                try:
                    t = timeit(fn, iters=6, warm=1)
                finally:
                    ops.igemm_force(0, 0, 0, 0)
                res.append((t, bm, bn, sp, blocks, kg))
    for t, bm, bn, sp, blocks, kg in res:
        print(f"CSV,{name},{M},{N},{nk},{bm},{bn},{sp},{kg},{t*1e6:.1f}")
    print(f"CSV,{name},{M},{N},{nk},0,0,0,0,{t_auto*1e6:.1f}")
    res.sort()
    best = res[0]
    line = "  ".join(f"{bm}x{bn}/s{sp}/g{kg}:{t*1e6:.0f}us" for t, bm, bn, sp, _, kg in res[:5])
    print(f"{name:34s} auto {t_auto*1e6:7.1f} us {flops/t_auto/1e12:6.1f} TF | best {best[0]*1e6:7.1f} us "
          f"{flops/best[0]/1e12:6.1f} TF ({best[1]}x{best[2]} s{best[3]} g{best[5]} blocks={best[4]}) | {line}", flush=True)


def main():
    torch.manual_seed(0)
    print(f"device: {torch.cuda.get_device_name(0)}")
    for (L, K, N) in [(1024, 256, 2048), (1024, 256, 768), (256, 384, 3072), (1024, 256, 256), (64, 640, 640),
                      (64, 2560, 640), (1024, 1024, 256), (256, 384, 384), (64, 640, 5120), (256, 384, 1152),
                      (256, 1536, 384), (64, 640, 1920), (1, 512, 640)]:
        M = B * L
        x = torch.randn(M, K, device="cuda")
        pw = ops.pack_conv(torch.randn(N, K) / math.sqrt(K), torch.randn(N))
        sweep(f"linear M={M} K={K} N={N}", lambda: ops.linear(x, pw), 2.0 * M * K * N, M, N, K // 32)
    for (H, W, Ci, Co, pre) in [(32, 2, 640, 640, 1), (64, 4, 384, 384, 1), (256, 16, 128, 128, 1),
                                (128, 8, 256, 256, 1), (32, 2, 1280, 640, 1), (256, 16, 128, 128, 0),
                                (128, 8, 256, 256, 0)]:
        x = torch.randn(B, H, W, Ci, device="cuda")
        pw = ops.pack_conv(torch.randn(Co, Ci, 3, 3) / math.sqrt(Ci * 9), torch.randn(Co))
        sc = torch.rand(B, Ci, device="cuda") + 0.5
        sh = torch.randn(B, Ci, device="cuda")
        if pre:
            fn = lambda: ops.conv(x, pw, pad=(1, 1), pre=(sc, sh), pre_act=ACT_SILU)
        else:
            fn = lambda: ops.conv(x, pw, pad=(1, 1))
        M = B * H * W
        sweep(f"conv3x3 {H}x{W} {Ci}->{Co} pre={pre}", fn, 2.0 * M * Ci * 9 * Co, M, Co, Ci * 9 // 32)


if __name__ == "__main__":
    main()
