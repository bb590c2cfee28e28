"""Error of the three matrix-core paths against an fp64 reference of the same contraction: fp32 MFMA (exact fp32 products),
"bf16x6" (exact 3-part operand split, 6 partial products) and "bf16x3" ((hi, mid) rounded to nearest, 3 partial products),
plus a plain single-pass bf16 product for scale.  Operands ~ N(0, 1) activations x N(0, 1/K) weights (what the layers see),
and a hard case with a large common offset.  max = max|err| / max|ref|, rms = rms(err) / rms(ref)."""
import math
import sys

import torch

sys.path.insert(0, ".")
from audioldm2_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(0)


def errs(y, ref):
    e = (y.double().cpu() - ref)
    return float(e.abs().max() / ref.abs().max()), float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


for name, M, K, N, off in [("K=256", 4096, 256, 256, 0.0), ("K=1152", 4096, 1152, 128, 0.0), ("K=5760", 1024, 5760, 640, 0.0),
                           ("K=1152 offset 3", 4096, 1152, 128, 3.0)]:
    x = torch.randn(M, K, generator=g) + off
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    ref = x.double() @ w.double().t()
    line = f"{name:18s}"
    xc = x.cuda().view(1, 1, M, K)
    for mode in ("f32", "bf16x6", "bf16x3"):
        ops.set_mma(mode)
        pw = ops.pack_conv(w)
        if mode == "f32":
            y = ops.conv(xc, pw)
        else:
            y = ops.conv(ops.split_rows(xc), pw)
        mx, rm = errs(y.view(M, N), ref)
        line += f" | {mode}: max {mx:.2e} rms {rm:.2e}"
    yb = (x.cuda().bfloat16() @ w.cuda().bfloat16().t()).float()
    mx, rm = errs(yb, ref)
    line += f" | plain bf16: max {mx:.2e} rms {rm:.2e}"
    print(line, flush=True)
ops.set_mma("bf16x6")
