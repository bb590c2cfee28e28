"""Halo-patch 3x3 convolution kernel (csrc/igemm_dma_halo.h) vs the tuned choice of the existing DMA-fed kernels on the sampling
path's 3x3 / stride-1 convolutions, same box, same process: agreement (fp32 summation order) and time per launch, HIP-graph timed.
Usage (GPU box): python tools/halo_probe.py [bf16x6|bf16x3] [--vae]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "bf16x6"
VAE = "--vae" in sys.argv
QUICK = "--quick" in sys.argv   # ablation builds: three shapes, two forms, no result check
ops.set_mma(MODE)
NP = ops.split_parts()


def g(seed):
    return torch.Generator().manual_seed(seed)


def graph_time(fn, reps=10, replays=4):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


class Case:
    def __init__(self, name, B, H, W, C, N):
        self.name, self.B, self.H, self.W, self.C, self.N = name, B, H, W, C, N
        x = torch.randn(B, H, W, C, generator=g(1)).cuda()
        w = torch.randn(N, C, 3, 3, generator=g(2)) / math.sqrt(C * 9)
        self.pw = ops.pack_conv(w, torch.randn(N, generator=g(3)))
        self.xs = ops.split_rows(x)
        self.emb = torch.randn(B, N, generator=g(4)).cuda()
        self.flops = 2.0 * B * H * W * C * 9 * N

    def run(self, force=None):
        if force:
            ops.igemm_force(*force)
        try:
            return ops.conv(self.xs, self.pw, pad=(1, 1), rowbias=self.emb)
        finally:
            if force:
                ops.igemm_force(0, 0, 0)


def main():
    if VAE:
        cases = [Case("vae 8x256x16   512->512", 8, 256, 16, 512, 512), Case("vae 8x512x32   512->512", 8, 512, 32, 512, 512),
                 Case("vae 8x512x32   256->256", 8, 512, 32, 256, 256), Case("vae 8x1024x64  256->256", 8, 1024, 64, 256, 256),
                 Case("vae 8x1024x64  128->128", 8, 1024, 64, 128, 128)]
    else:
        cases = [Case("L0 16x256x16  128->128", 16, 256, 16, 128, 128), Case("L0  8x256x16  128->128", 8, 256, 16, 128, 128),
                 Case("L0 16x256x16  256->128", 16, 256, 16, 256, 128), Case("L0 16x256x16  384->128", 16, 256, 16, 384, 128),
                 Case("L1 16x128x8   256->256", 16, 128, 8, 256, 256), Case("L1  8x128x8   128->256", 8, 128, 8, 128, 256),
                 Case("L1 16x128x8   512->256", 16, 128, 8, 512, 256), Case("L1 16x128x8   640->256", 16, 128, 8, 640, 256),
                 Case("L2 16x64x4    384->384", 16, 64, 4, 384, 384), Case("L2 16x64x4    768->384", 16, 64, 4, 768, 384),
                 Case("L2 16x64x4   1024->384", 16, 64, 4, 1024, 384), Case("L3 16x32x2    640->640", 16, 32, 2, 640, 640),
                 Case("L3 16x32x2   1280->640", 16, 32, 2, 1280, 640)]
    if QUICK:
        cases = [c for c in cases if c.name in ("L0 16x256x16  128->128", "L0 16x256x16  256->128", "L1 16x128x8   256->256")]
    forms = [(256, 128, 402), (128, 128, 402), (128, 128, 403), (128, 128, 404), (128, 128, 412), (128, 128, 413)]
    if NP == 2:
        forms = [(256, 128, 402), (256, 128, 403), (128, 128, 403), (128, 128, 404), (128, 128, 413)]
    if QUICK:
        forms = [(256, 128, 402), (128, 128, 413)]
    print(f"# mode {MODE} ({NP}-part images); us per launch, HIP-graph timed; TF = fp32-equivalent TFLOP/s; sN = split-K N", flush=True)
    for c in cases:
        t_auto = graph_time(lambda: c.run())
        y_ref = c.run()
        line = f"{c.name:26s} tuned {t_auto:7.1f} ({c.flops / t_auto / 1e6:5.0f} TF) |"
        best = (t_auto, "tuned")
        cpb = c.C // 32
        for bm, bn, st in forms:
            for sp in (1, 2, 3, 4):
                if sp > 1 and (cpb % sp != 0 or c.B * c.H * c.W > 8192):
                    continue
                try:
                    y = c.run((bm, bn, sp, 0, st))
                except RuntimeError:
                    continue
                err = float((y.double() - y_ref.double()).abs().max() / y_ref.double().abs().max())
                t = graph_time(lambda: c.run((bm, bn, sp, 0, st)))
                tag = f"{bm}/{st}" + (f"s{sp}" if sp > 1 else "")
                line += f" {tag} {t:6.1f}{'' if QUICK or err < (2.5e-6 if NP == 3 else 1e-5) else f' !err {err:.1e}'}"
                if t < best[0]:
                    best = (t, tag)
        print(line + f" -> best {best[1]} {best[0]:.1f} ({c.flops / best[0] / 1e6:.0f} TF, {100 * (best[0] / t_auto - 1):+.0f} %)", flush=True)


if __name__ == "__main__":
    main()
