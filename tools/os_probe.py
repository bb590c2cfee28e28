"""Operand-stationary DMA GEMM (csrc/igemm_dma_os.h) vs the tuned igemm_dma_kernel choice on the UNet's K = C projections, same
box, same process: agreement with the classic kernel on the 64x128 tile (fp32 rounding), and time per launch, HIP-graph timed (R
launches per replay).  Usage (GPU box): python tools/os_probe.py [bf16x6|bf16x3] [--rows]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "bf16x6"
ROWS = "--rows" in sys.argv
ops.set_mma(MODE)
NP = ops.split_parts()


def g(seed):
    return torch.Generator().manual_seed(seed)


def graph_time(fn, reps=20, replays=4):
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
    def __init__(self, name, kind, M, K, N, res=False, bias=True, split_out=None, heads=0, L=0):
        self.name, self.kind, self.M, self.K, self.N, self.split_out, self.heads, self.L = name, kind, M, K, N, split_out, heads, L
        x = torch.randn(1, M, K, generator=g(1)).cuda()
        w = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
        b = torch.randn(N, generator=g(3)) if bias else None
        self.pw = ops.pack_geglu(w, b) if kind == "geglu" else ops.pack_conv(w, b)
        self.xs = ops.split_rows(x)
        if kind == "qkv":
            self.xs = self.xs.view(M // L, L, K)
        self.res = torch.randn((1, M, N), generator=g(4)).cuda() if res else None
        self.flops = 2.0 * M * K * N

    def run(self, force=None):
        if force:
            ops.igemm_force(force[0], force[1], 1, 0, force[2])
        try:
            if self.kind == "geglu":
                return ops.linear_geglu(self.xs, self.pw, split_out=self.split_out)
            if self.kind == "qkv":
                return ops.linear_qkv(self.xs, self.pw, self.heads, self.L)
            return ops.linear(self.xs, self.pw, res=self.res, split_out=self.split_out)
        finally:
            if force:
                ops.igemm_force(0, 0, 0)


def close(a, b):
    """fp32 outputs agree to 2e-6 max-norm; raw part images are compared after summing their parts (any layout: the part axis is
    the one of size split_parts next to 32-wide rows — decode every int16 as a bf16 and compare the sums over ALL elements' blocks)."""
    if a.dtype == torch.int16:
        fa = (a.to(torch.int32) << 16).view(torch.float32).double()
        fb = (b.to(torch.int32) << 16).view(torch.float32).double()
        return float((fa.sum() - fb.sum()).abs()) <= 1e-3 * float(fb.abs().sum()) and \
            float((fa - fb).abs().max()) <= 2.0 ** -7 * float(fb.abs().max())   # hi parts equal up to an ulp of bf16
    return float((a.double() - b.double()).abs().max()) <= 2e-6 * float(b.double().abs().max())


def flat(y):
    ys = y if isinstance(y, tuple) else (y,)
    return [t.data if isinstance(t, ops.SplitT) else t for t in ys]


def main():
    R = 16
    cases = [
        Case("L1 geglu 16384x256->2x1024 (split out)", "geglu", R * 1024, 256, 2048, split_out="only"),
        Case("L1 qkv   16384x256->768 (QKV epilogue)", "qkv", R * 1024, 256, 768, bias=False, heads=8, L=1024),
        Case("L1 proj  16384x256->256 +bias +res", "linear", R * 1024, 256, 256, res=True),
        Case("L1 q2    16384x256->256", "linear", R * 1024, 256, 256, bias=False),
        Case("L1 pout  16384x256->256 +bias +res, split in", "linear", R * 1024, 256, 256, res=True),
        Case("L2 geglu 4096x384->2x1536 (split out)", "geglu", R * 256, 384, 3072, split_out="only"),
        Case("L2 qkv   4096x384->1152 (QKV epilogue)", "qkv", R * 256, 384, 1152, bias=False, heads=12, L=256),
        Case("L2 proj  4096x384->384 +bias +res", "linear", R * 256, 384, 384, res=True),
    ]
    print(f"# mode {MODE} ({NP}-part images); us per launch, HIP-graph timed; TF = fp32-equivalent TFLOP/s", flush=True)
    for c in cases:
        depths = ([2, 3] if c.K == 256 else [2]) if NP == 3 else ([2, 3, 4] if c.K == 256 else [2, 3])
        t_auto = graph_time(lambda: c.run())
        y_old = flat(c.run((64, 128, 2)))
        line = f"{c.name:48s} auto {t_auto:6.1f} ({c.flops / t_auto / 1e6:5.0f} TF) |"
        best = (t_auto, "auto")
        for st in depths:
            y_os = flat(c.run((32, 128, 300 + st)))
            same = all(close(a, b) for a, b in zip(y_os, y_old))
            t = graph_time(lambda: c.run((32, 128, 300 + st)))
            line += f" os{st} {t:6.1f} ({c.flops / t / 1e6:5.0f} TF){'' if same else ' !~classic'}"
            if t < best[0]:
                best = (t, f"os{st}")
        if ROWS:
            st = depths[-1]
            for rows in (64, 128, 256, 512, 1024, 2048):
                os.environ["ALDM_OS_ROWS"] = str(rows)
                t = graph_time(lambda: c.run((32, 128, 300 + st)))
                line += f" r{rows} {t:6.1f}"
            os.environ.pop("ALDM_OS_ROWS")
        print(line + f" -> best {best[1]} {best[0]:.1f}", flush=True)


if __name__ == "__main__":
    main()
