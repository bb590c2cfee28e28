"""Per-launch time of the pre-split self-attention kernel at the UNet's geometries (16 samples per pass), HIP-graph timed, for
the library named by $ALDM_LIB_PATH and the kernel selected by $ALDM_ATTN_SCHED (0 = the round-3/4 pipelined kernel, unset / 1 = the
round-5 re-scheduled exact-max loop, 2 = the opt-in one-pass fixed-reference loop) — one process per arm (both switches are read
once).  Also checks the result against fp64 and against the fp32-K/V path (bitwise for 0 / 1; to fp32 rounding for 2).
Usage: [ALDM_LIB_PATH=...] [ALDM_ATTN_SCHED=0|2] [ALDM_MMA=bf16x3] python tools/attn_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402

GEOMS = [(16, 8, 1024), (16, 12, 256), (16, 20, 64)]   # (samples, heads, tokens) of levels 1-3
tag = f"{os.path.basename(os.environ.get('ALDM_LIB_PATH', 'libaldm_hip.so'))} sched={os.environ.get('ALDM_ATTN_SCHED', '1')} {ops.MMA_MODE}"


def timed(fn, reps=50):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


for B, H, L in GEOMS:
    C = H * 32
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, L, C, generator=gen)
    w = torch.randn(3 * C, C, generator=gen) / C ** 0.5
    xs = ops.split_rows(x.cuda())
    pw = ops.pack_conv(w, None)
    q, k_img, vt_img = ops.linear_qkv(xs, pw, H, L)
    out = ops.attention_presplit(q, k_img, vt_img, H)
    # the same projection through the plain path -> fp32 k / v -> the in-kernel-split attention: bitwise the same result
    qkv = ops.linear(xs, pw)
    o_ref = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], H)
    bit = bool(torch.equal(out, o_ref))
    dif = float((out.double() - o_ref.double()).abs().max() / o_ref.double().abs().max())
    sh = lambda t: t.double().cpu().view(B, L, H, 32).transpose(1, 2)
    qd, kd, vd = (sh(qkv[..., i * C:(i + 1) * C].contiguous()) for i in range(3))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) / 32 ** 0.5, -1) @ vd).transpose(1, 2).reshape(B, L, C)
    err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
    us = timed(lambda: ops.attention_presplit(q, k_img, vt_img, H))
    fl = 4.0 * B * H * L * L * 32
    print(f"{tag}: {B} x {H} heads x {L} x {L}: {us:7.2f} us  {fl / us * 1e-6:6.1f} TFLOP/s  err vs fp64 {err:.2e}  vs fp32-KV path: bitwise {bit}, max diff {dif:.1e}",
          flush=True)
