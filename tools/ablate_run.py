"""Time a few forced-tile bf16-split launches under the library named by $ALDM_LIB_PATH (ablation builds from
tools/gpu/build_ablate.sh).  Usage: ALDM_LIB_PATH=tools/gpu/libaldm_abl3.so python tools/ablate_run.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402
from audioldm2_amd.ops import ACT_SILU  # noqa: E402

tag = os.path.basename(os.environ.get("ALDM_LIB_PATH", "libaldm_hip.so"))
B, reps = 16, 10
torch.manual_seed(0)
ops.set_mma("bf16x6")
x = torch.randn(B, 256, 16, 128, device="cuda")
pw = ops.pack_conv(torch.randn(128, 128, 3, 3) / 34.0)
sc = torch.rand(B, 128, device="cuda") + 0.5
sh = torch.randn(B, 128, device="cuda")
xl = torch.randn(B * 1024, 1024, device="cuda")
pl = ops.pack_conv(torch.randn(256, 1024) / 32.0)
cases = [("conv128 plain", lambda: ops.conv(x, pw, pad=(1, 1))),
         ("conv128 gn+silu", lambda: ops.conv(x, pw, pad=(1, 1), pre=(sc, sh), pre_act=ACT_SILU)),
         ("linear 16384x1024x256", lambda: ops.linear(xl, pl))]
out = []
for name, fn in cases:
    for bm, bn in ((128, 128), (64, 128)):
        ops.igemm_force(bm, bn, 1)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(f"{name} {bm}x{bn} {e0.elapsed_time(e1) / reps * 1e3:7.1f}")
        ops.igemm_force(0, 0, 0)
print(f"{tag:22s} | " + " | ".join(out), flush=True)
