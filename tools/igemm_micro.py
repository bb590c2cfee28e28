"""A few representative igemm launches repeated (for rocprofv3 --pmc runs).
Usage: python tools/igemm_micro.py [iters]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402
from audioldm2_amd.ops import ACT_SILU  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = 16
torch.manual_seed(0)
for (H, W, Ci, Co, pre) in [(256, 16, 128, 128, 0), (256, 16, 128, 128, 1), (128, 8, 256, 256, 1),
                            (64, 4, 384, 384, 1), (32, 2, 640, 640, 1)]:
    x = torch.randn(B, H, W, Ci, device="cuda")
    pw = ops.pack_conv(torch.randn(Co, Ci, 3, 3) / math.sqrt(Ci * 9))
    sc = torch.rand(B, Ci, device="cuda") + 0.5
    sh = torch.randn(B, Ci, device="cuda")
    for _ in range(iters):
        if pre:
            ops.conv(x, pw, pad=(1, 1), pre=(sc, sh), pre_act=ACT_SILU)
        else:
            ops.conv(x, pw, pad=(1, 1))
for (L, K, N) in [(1024, 256, 768), (1024, 256, 2048), (1024, 1024, 256), (256, 384, 3072), (64, 640, 5120),
                  (64, 2560, 640)]:
    x = torch.randn(B * L, K, device="cuda")
    pw = ops.pack_conv(torch.randn(N, K) / math.sqrt(K))
    for _ in range(iters):
        ops.linear(x, pw)
torch.cuda.synchronize()
print("done")
