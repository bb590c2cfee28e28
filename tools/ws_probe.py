"""Persistent wave-specialised DMA GEMM (csrc/igemm_dma_ws.h) vs igemm_dma_kernel on the UNet's GEMM shapes, same box, same
process: (1) results — bitwise against the non-persistent kernel on the same tile (same K order, same epilogue order) and
max-norm relative against fp64 on the CPU; (2) time per launch, HIP-graph timed (R launches per replay: the ctypes launch path
would hide every short kernel).  Usage (GPU box): python tools/ws_probe.py [bf16x3|bf16x6] [--quick]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "bf16x3"
QUICK = "--quick" in sys.argv
ops.set_mma(MODE)
NP = ops.split_parts()
WS_CFGS = {2: [(64, 128, 2), (64, 128, 3), (64, 128, 4), (128, 64, 2), (128, 64, 3), (64, 64, 3), (64, 64, 4), (64, 64, 6)],
           3: [(64, 128, 2), (64, 128, 3), (128, 64, 2), (128, 64, 3), (64, 64, 2), (64, 64, 3), (64, 64, 4)]}[NP]
OLD_DEFAULT_ST = {(64, 128): 4, (128, 64): 4, (64, 64): 3, (128, 128): 2}
LW_CFGS = {2: [(128, 128, 2), (128, 128, 4), (64, 128, 2), (64, 128, 3), (64, 128, 4), (128, 64, 2), (128, 64, 3), (128, 64, 4),
               (64, 64, 2), (64, 64, 3), (64, 64, 4)],
           3: [(128, 128, 2), (128, 128, 3), (64, 128, 2), (64, 128, 4), (128, 64, 2), (128, 64, 4), (64, 64, 2), (64, 64, 3)]}[NP]
ONLY = os.environ.get("WS_PROBE", "ws,lw").split(",")


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
    """kind: linear | geglu | conv3.  Builds operands once; run(force) launches with a forced (bm, bn, stages) or auto."""

    def __init__(self, name, kind, M, K, N, res=False, rowbias=False, split_out=None, hw=None):
        self.name, self.kind, self.M, self.K, self.N = name, kind, M, K, N
        self.split_out = split_out
        if kind == "conv3":
            B, H, W = hw
            C = K // 9
            self.x = torch.randn(B, H, W, C, generator=g(1)).cuda()
            w = torch.randn(N, C, 3, 3, generator=g(2)) / math.sqrt(K)
            self.w4 = w
            self.pw = ops.pack_conv(w, torch.randn(N, generator=g(3)))
            self.xs = ops.split_rows(self.x)
            oshape = (B, H, W, N)
            self.B = B
        else:
            self.x = torch.randn(1, M, K, generator=g(1)).cuda()
            nfull = N
            w = torch.randn(nfull, K, generator=g(2)) / math.sqrt(K)
            b = torch.randn(nfull, generator=g(3))
            self.w2, self.b = w, b
            self.pw = ops.pack_geglu(w, b) if kind == "geglu" else ops.pack_conv(w, b)
            self.xs = ops.split_rows(self.x)
            oshape = (1, M, N // 2 if kind == "geglu" else N)
            self.B = 1
        self.res = torch.randn(oshape, generator=g(4)).cuda() if res else None
        self.rowbias = torch.randn(self.B, N, generator=g(5)).cuda() if rowbias else None

    def run(self, force=None):
        if force:
            ops.igemm_force(force[0], force[1], 1, 0, force[2])
        try:
            if self.kind == "geglu":
                return ops.linear_geglu(self.xs, self.pw, split_out=self.split_out)
            if self.kind == "conv3":
                return ops.conv(self.xs, self.pw, pad=(1, 1), res=self.res, rowbias=self.rowbias, split_out=self.split_out)
            return ops.linear(self.xs, self.pw, res=self.res, split_out=self.split_out)
        finally:
            if force:
                ops.igemm_force(0, 0, 0)

    def reference(self):
        x = self.xs.float().double().cpu()   # the operand the kernels see (2-part images are already rounded)
        if self.kind == "conv3":
            y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), self.w4.double(), self.pw.bias.double().cpu(), padding=1)
            y = y.permute(0, 2, 3, 1)
            if self.rowbias is not None:
                y = y + self.rowbias.double().cpu()[:, None, None, :]
        elif self.kind == "geglu":
            h = x @ self.w2.double().t() + self.b.double()
            a, gate = h.chunk(2, -1)
            y = a * torch.nn.functional.gelu(gate)
        else:
            y = x @ self.w2.double().t() + self.b.double()
        if self.res is not None:
            y = y + self.res.double().cpu()
        return y


def as_float(y):
    if isinstance(y, tuple):
        y = y[0]
    return y.float() if isinstance(y, ops.SplitT) else y


def main():
    R = 16   # samples per pass (CFG batch of 8 prompts)
    cases = [
        Case("L1 geglu 16384x256->2x1024 (split out)", "geglu", R * 1024, 256, 2048, split_out="only"),
        Case("L1 qkv 16384x256->768", "linear", R * 1024, 256, 768),
        Case("L1 proj 16384x256->256 +res", "linear", R * 1024, 256, 256, res=True),
        Case("L1 ffout 16384x1024->256 +res", "linear", R * 1024, 1024, 256, res=True),
        Case("L2 geglu 4096x384->2x1536 (split out)", "geglu", R * 256, 384, 3072, split_out="only"),
        Case("L2 qkv 4096x384->1152", "linear", R * 256, 384, 1152),
        Case("L2 proj 4096x384->384 +res", "linear", R * 256, 384, 384, res=True),
        Case("L2 ffout 4096x1536->384 +res", "linear", R * 256, 1536, 384, res=True),
        Case("L3 geglu 1024x640->2x2560 (split out)", "geglu", R * 64, 640, 5120, split_out="only"),
        Case("L3 proj 1024x640->640 +res", "linear", R * 64, 640, 640, res=True),
        Case("L0 conv3x3 128->128 @256x16 +rowbias", "conv3", R * 4096, 1152, 128, rowbias=True, hw=(R, 256, 16)),
        Case("L1 conv3x3 256->256 @128x8 +res (also split)", "conv3", R * 1024, 2304, 256, res=True, split_out="also",
             hw=(R, 128, 8)),
    ]
    if QUICK:
        cases = cases[:4]
    print(f"# mode {MODE} ({NP}-part images); times: us per launch, HIP-graph timed", flush=True)
    for c in cases:
        ref = c.reference()
        y_auto = as_float(c.run())
        e_auto = float((y_auto.double().cpu() - ref).abs().max() / ref.abs().max())
        t_auto = graph_time(lambda: c.run())
        line = f"{c.name:52s} auto {t_auto:7.1f} (err {e_auto:.1e}) |"
        best = (t_auto, "auto")
        for bm, bn, st in (LW_CFGS if "lw" in ONLY else []):
            if c.kind == "geglu" and bn != 128:
                continue
            try:
                y_lw = c.run((bm, bn, 200 + st))
            except RuntimeError as e:
                line += f" lw{bm}x{bn}s{st} n/a({str(e)[:40]})"
                continue
            y_old = c.run((bm, bn, OLD_DEFAULT_ST[(bm, bn)]))
            same = torch.equal(as_float(y_lw), as_float(y_old))
            e_lw = float((as_float(y_lw).double().cpu() - ref).abs().max() / ref.abs().max())
            t_lw = graph_time(lambda: c.run((bm, bn, 200 + st)))
            line += f" lw{bm}x{bn}s{st} {t_lw:6.1f}{'' if same else ' !=old'}{'' if e_lw < 5e-5 else f' ERR {e_lw:.1e}'}"
            if t_lw < best[0]:
                best = (t_lw, f"lw{bm}x{bn}s{st}")
        for bm, bn, st in (WS_CFGS if "ws" in ONLY else []):
            if c.M % bm or c.N % bn or (c.kind == "geglu" and bn != 128):
                continue
            if c.kind == "conv3" and c.rowbias is not None and (c.M // c.B) % bm:
                continue
            try:
                y_ws = c.run((bm, bn, 100 + st))
            except RuntimeError as e:
                line += f" ws{bm}x{bn}s{st} n/a({str(e)[:40]})"
                continue
            y_old = c.run((bm, bn, OLD_DEFAULT_ST[(bm, bn)]))
            same = torch.equal(as_float(y_ws), as_float(y_old))
            if isinstance(y_ws, tuple):
                same = same and torch.equal(y_ws[1].data, y_old[1].data)
            e_ws = float((as_float(y_ws).double().cpu() - ref).abs().max() / ref.abs().max())
            t_ws = graph_time(lambda: c.run((bm, bn, 100 + st)))
            line += f" ws{bm}x{bn}s{st} {t_ws:6.1f}{'' if same else ' !=old'}{'' if e_ws < 5e-5 else f' ERR {e_ws:.1e}'}"
            if t_ws < best[0]:
                best = (t_ws, f"ws{bm}x{bn}s{st}")
        print(line + f" -> best {best[1]} {best[0]:.1f}", flush=True)


if __name__ == "__main__":
    main()
