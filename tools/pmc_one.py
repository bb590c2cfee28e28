"""One igemm shape, forced tile, repeated (for rocprofv3 --pmc passes).
Usage: python tools/pmc_one.py <f32|bf16x6> <bm> <bn> [gn]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402
from audioldm2_amd.ops import ACT_SILU  # noqa: E402

mode, bm, bn = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
gn = len(sys.argv) > 4
ops.set_mma(mode)
torch.manual_seed(0)
B = 16
x = torch.randn(B, 256, 16, 128, device="cuda")
pw = ops.pack_conv(torch.randn(128, 128, 3, 3) / 34.0)
sc = torch.rand(B, 128, device="cuda") + 0.5
sh = torch.randn(B, 128, device="cuda")
kw = dict(pre=(sc, sh), pre_act=ACT_SILU) if gn else {}
ops.igemm_force(bm, bn, 1)
for _ in range(10):
    ops.conv(x, pw, pad=(1, 1), **kw)
torch.cuda.synchronize()
print("done")
