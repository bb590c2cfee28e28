"""fp32-MFMA vs bf16-split (BF16x6) igemm on representative UNet shapes: time per launch and max error of both
against an fp64 reference of the same product.  Usage: python tools/mma_ab.py [reps]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402
from audioldm2_amd.ops import ACT_SILU  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = 16
torch.manual_seed(0)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, y


def run(name, fn, flops, ref=None):
    row = [f"{name:44s}"]
    ops.set_mma("f32")
    t, y32 = timed(fn)
    row.append(f"f32 {t:8.1f} us {flops / t / 1e6:6.1f} TF/s")
    ops.set_mma("bf16x6")
    t, yb = timed(fn)
    row.append(f"| bx auto {t:8.1f} us {flops / t / 1e6:6.1f} TF/s")
    for bm, bn in ((128, 128), (64, 128), (128, 64), (64, 64)):
        ops.igemm_force(bm, bn, 1)
        try:
            t, _ = timed(fn)
            row.append(f"{bm}x{bn} {t:7.1f}")
        except RuntimeError as e:
            row.append(f"{bm}x{bn}    n/a ")
        finally:
            ops.igemm_force(0, 0, 0)
    if ref is not None:
        r = ref()
        den = r.abs().max().item()
        row.append(f"| err f32 {(y32.double() - r).abs().max().item() / den:.2e} bx {(yb.double() - r).abs().max().item() / den:.2e}")
    print(" ".join(row), flush=True)
    ops.set_mma("f32")


for (H, W, Ci, Co, pre) in [(256, 16, 128, 128, 0), (256, 16, 128, 128, 1), (256, 16, 256, 128, 1), (128, 8, 256, 256, 1),
                            (64, 4, 384, 384, 1), (32, 2, 640, 640, 1)]:
    x = torch.randn(B, H, W, Ci, device="cuda")
    w = torch.randn(Co, Ci, 3, 3) / math.sqrt(Ci * 9)
    pw = ops.pack_conv(w)
    sc = torch.rand(B, Ci, device="cuda") + 0.5
    sh = torch.randn(B, Ci, device="cuda")
    kw = dict(pre=(sc, sh), pre_act=ACT_SILU) if pre else {}
    ref = None
    if (H, Ci) in ((256, 128), (32, 640)):
        def ref(x=x, w=w, sc=sc, sh=sh, pre=pre):
            xd = x.double()
            if pre:
                xd = torch.nn.functional.silu(xd * sc.double()[:, None, None, :] + sh.double()[:, None, None, :])
            return torch.nn.functional.conv2d(xd.permute(0, 3, 1, 2), w.double().cuda(), padding=1).permute(0, 2, 3, 1)
    run(f"conv3x3 {Ci}->{Co} @{H}x{W} {'gn+silu' if pre else 'plain'}", lambda: ops.conv(x, pw, pad=(1, 1), **kw),
        2.0 * B * H * W * Co * Ci * 9, ref)
for (L, K, N) in [(4096, 128, 384), (1024, 256, 768), (1024, 256, 2048), (1024, 1024, 256), (256, 384, 3072), (256, 1536, 384),
                  (64, 640, 5120), (64, 2560, 640)]:
    x = torch.randn(B * L, K, device="cuda")
    w = torch.randn(N, K) / math.sqrt(K)
    pw = ops.pack_conv(w)
    ref = (lambda x=x, w=w: x.double() @ w.double().cuda().t()) if L == 1024 else None
    run(f"linear M={B * L} K={K} N={N}", lambda: ops.linear(x, pw), 2.0 * B * L * K * N, ref)
# GEGLU projection
x = torch.randn(B * 1024, 256, device="cuda")
pw = ops.pack_geglu(torch.randn(2048, 256) / 16, torch.randn(2048))
run("geglu M=16384 K=256 N=2048", lambda: ops.linear_geglu(x, pw), 2.0 * B * 1024 * 256 * 2048)
print("done")
