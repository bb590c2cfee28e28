"""Micro-benchmark of the hot kernels on the UNet's real shapes (prints TFLOP/s per shape).
Usage (GPU box): python tools/bench_ops.py [batch]
"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    print(f"device: {torch.cuda.get_device_name(0)}  effective batch (incl. CFG) = {B}")
    rows = []
    # 3x3 convs of the four UNet levels (H, W, Cin, Cout)
    for (H, W, Ci, Co) in [(256, 16, 128, 128), (256, 16, 256, 128), (128, 8, 256, 256),
                           (128, 8, 512, 256), (64, 4, 384, 384), (64, 4, 768, 384),
                           (32, 2, 640, 640), (32, 2, 1280, 640)]:
        x = torch.randn(B, H, W, Ci, device="cuda")
        w = torch.randn(Co, Ci, 3, 3) / math.sqrt(Ci * 9)
        pw = ops.pack_conv(w)
        t = timeit(lambda: ops.conv(x, pw, pad=(1, 1)))
        fl = 2.0 * B * H * W * Ci * 9 * Co
        rows.append((f"conv3x3 {H}x{W} {Ci}->{Co}", t, fl))
    # linears (rows = tokens)
    for (L, K, N) in [(1024, 256, 768), (1024, 256, 2048), (1024, 1024, 256), (256, 384, 3072),
                      (256, 1536, 384), (64, 640, 5120), (64, 2560, 640)]:
        x = torch.randn(B * L, K, device="cuda")
        pw = ops.pack_conv(torch.randn(N, K) / math.sqrt(K))
        t = timeit(lambda: ops.linear(x, pw))
        rows.append((f"linear M={B*L} {K}->{N}", t, 2.0 * B * L * K * N))
    # attention
    for (h, L, Lk) in [(8, 1024, 1024), (12, 256, 256), (20, 64, 64), (8, 1024, 32)]:
        q = torch.randn(B, L, h * 32, device="cuda")
        k = torch.randn(B, Lk, h * 32, device="cuda")
        v = torch.randn(B, Lk, h * 32, device="cuda")
        t = timeit(lambda: ops.attention(q, k, v, h))
        rows.append((f"attn h={h} Lq={L} Lk={Lk}", t, 4.0 * B * h * L * Lk * 32))
    # memory-bound
    x = torch.randn(B, 256, 16, 128, device="cuda")
    gam = torch.ones(128, device="cuda")
    t = timeit(lambda: ops.gn_stats(x, gam, gam))
    print(f"gn_stats [B,4096,128]: {t*1e6:8.1f} us  {x.numel()*4/t/1e9:8.1f} GB/s")
    x2 = torch.randn(B * 1024, 256, device="cuda")
    g2 = torch.ones(256, device="cuda")
    t = timeit(lambda: ops.layernorm(x2, g2, g2))
    print(f"layernorm [B*1024,256]: {t*1e6:8.1f} us  {2*x2.numel()*4/t/1e9:8.1f} GB/s")
    for name, t, fl in rows:
        print(f"{name:34s} {t*1e6:9.1f} us  {fl/t/1e12:7.2f} TFLOP/s")


if __name__ == "__main__":
    main()
