"""Same-process A/B of the register-staged bf16-split igemm (round 1) against the DMA-fed kernel over pre-split operands
(csrc/igemm_dma.h) on the UNet's dominant shapes at 16 samples/pass: us per launch, fp32-equivalent TFLOP/s, fraction of the
416.7 TFLOP/s bf16x6 ceiling; the DMA kernel per (tile, ring depth, split-K).  Also times the producers (split_rows)."""
import itertools
import math
import sys

import torch

sys.path.insert(0, ".")
from audioldm2_amd import ops  # noqa: E402

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(fn, iters=ITERS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ops.set_mma("bf16x6")
    g = torch.Generator().manual_seed(0)
    shapes = [  # name, B, H, W, C, N, k
        ("conv3x3 128->128 @256x16", 16, 256, 16, 128, 128, 3),
        ("conv3x3 256->128 @256x16", 16, 256, 16, 256, 128, 3),
        ("conv3x3 256->256 @128x8", 16, 128, 8, 256, 256, 3),
        ("conv3x3 384->384 @64x4", 16, 64, 4, 384, 384, 3),
        ("conv3x3 640->640 @32x2", 16, 32, 2, 640, 640, 3),
        ("linear 16384x256->768", 1, 1, 16384, 256, 768, 1),
        ("linear 16384x256->256", 1, 1, 16384, 256, 256, 1),
        ("linear 16384x1024->256", 1, 1, 16384, 1024, 256, 1),
        ("linear 4096x384->1152", 1, 1, 4096, 384, 1152, 1),
        ("linear 4096x384->384", 1, 1, 4096, 384, 384, 1),
        ("linear 4096x1536->384", 1, 1, 4096, 1536, 384, 1),
        ("linear 1024x640->640", 1, 1, 1024, 640, 640, 1),
        ("linear 1024x640->1920", 1, 1, 1024, 640, 1920, 1),
        ("linear 1024x2560->640", 1, 1, 1024, 2560, 640, 1),
    ]
    cfgs = [(256, 128, 2), (128, 128, 3), (128, 128, 2), (64, 128, 4), (64, 128, 2), (128, 64, 4), (128, 64, 2), (64, 64, 3), (64, 64, 2)]
    for name, B, H, W, C, N, k in shapes:
        x = torch.randn(B, H, W, C, generator=g).cuda()
        w = torch.randn(N, C, k, k, generator=g) / math.sqrt(C * k * k)
        pw = ops.pack_conv(w, torch.randn(N, generator=g))
        pad = (k // 2, k // 2)
        M = B * H * W
        fl = 2.0 * M * N * C * k * k
        t_old = timeit(lambda: ops.conv(x, pw, pad=pad))
        xs = ops.split_rows(x)
        t_split = timeit(lambda: ops.split_rows(x))
        y_old = ops.conv(x, pw, pad=pad)
        line = f"{name:28s} old {t_old:7.1f} us {fl / t_old * 1e-6:6.1f} TF | split_rows {t_split:6.1f} us |"
        t_auto = timeit(lambda: ops.conv(xs, pw, pad=pad))
        y_new = ops.conv(xs, pw, pad=pad)
        err = float((y_new - y_old).abs().max() / y_old.abs().max())
        line += f" dma auto {t_auto:7.1f} us {fl / t_auto * 1e-6:6.1f} TF ({fl / t_auto * 1e-6 / 416.7:.2f}) err {err:.1e} |"
        best = None
        for (bm, bn, st), sp in itertools.product(cfgs, (1, 2, 4) if M <= 4096 else (1,)):
            nk = C * k * k // 32
            if sp > 1 and (nk // sp < 3 or M * N > 8 << 20):
                continue
            if (M + bm - 1) // bm * ((N + bn - 1) // bn) * sp > 4096 and bm * bn < 128 * 128:
                continue
            ops.igemm_force(bm, bn, sp, 0, st)
            try:
                t = timeit(lambda: ops.conv(xs, pw, pad=pad), iters=max(5, ITERS // 2))
            finally:
                ops.igemm_force(0, 0, 0)
            line += f" {bm}x{bn}s{st}k{sp} {t:6.1f}"
            if best is None or t < best[0]:
                best = (t, bm, bn, st, sp)
        line += f" | best {best[1]}x{best[2]} st{best[3]} k{best[4]} {best[0]:.1f} us {fl / best[0] * 1e-6:.1f} TF ({fl / best[0] * 1e-6 / 416.7:.2f})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
