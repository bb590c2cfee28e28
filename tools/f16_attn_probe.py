"""Per-launch time of the self-attention kernels on the operands the QKV epilogue writes, per UNet level (16 samples), HIP-graph timed:
the mode's kernel (f16x3: attention_d32_presplit_f16; bf16x6 / bf16x3: attention_d32_presplit) with the queries-per-wave choice the
library makes, or forced by $ALDM_ATTN_QT (read once per process).  Usage: ALDM_ATTN_QT=1|2 python tools/f16_attn_probe.py [mode]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
ops.set_mma(mode)
g = lambda s: torch.Generator().manual_seed(s)


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


for B, L, heads in ((16, 1024, 8), (8, 1024, 8), (16, 256, 12), (16, 64, 20)):
    C = heads * 32
    x = torch.randn(B, L, C, generator=g(1)).cuda()
    ga, be = torch.ones(C).cuda(), torch.zeros(C).cuda()
    w = torch.cat([torch.randn(C, C, generator=g(4 + i)) / math.sqrt(C) for i in range(3)], 0)
    pw = ops.pack_conv(w)
    n = ops.layernorm(x, ga, be, 1e-5, split_out="only")
    q, kimg, vtimg = ops.linear_qkv(n, pw, heads, L)
    t = graph_time(lambda: ops.attention_presplit(q, kimg, vtimg, heads, split_out="only"))
    fl = 4.0 * B * heads * L * L * 32
    print(f"{mode} ALDM_ATTN_QT={os.environ.get('ALDM_ATTN_QT', 'auto')}: {B} x {heads} heads x {L} x {L}: {t:7.2f} us  {fl / t / 1e6:6.1f} TFLOP/s", flush=True)
