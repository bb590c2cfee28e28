"""fp32-MFMA vs bf16x6 vs bf16x3 attention (aldm_attention_mma 1 / 2 / 3) on the UNet's attention shapes, same process;
20 launches per HIP-graph replay (an eager ctypes launch costs ~17 us of host time, more than the small shapes run)."""
import sys

import torch

sys.path.insert(0, ".")
from audioldm2_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(iters):
                keep = fn()
    torch.cuda.synchronize()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * iters)


g = torch.Generator().manual_seed(0)
for B, heads, Lq, Lk in [(16, 8, 1024, 1024), (16, 12, 256, 256), (16, 20, 64, 64), (16, 8, 1024, 32), (16, 8, 1024, 40), (16, 8, 1024, 8)]:
    qkv = torch.randn(B, Lq, 3 * heads * 32, generator=g).cuda()
    kv = torch.randn(B, Lk, 2 * heads * 32, generator=g).cuda()
    C = heads * 32
    q, k, v = (qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]) if Lq == Lk else (qkv[:, :, :C], kv[:, :, :C], kv[:, :, C:])
    fl = 4.0 * B * heads * Lq * Lk * 32
    res = {}
    mask = None
    if Lq != Lk:
        mask = torch.ones(B, Lk)
        mask[B // 2:, Lk - 8:] = 0
        mask = mask.cuda()
    ref = torch.nn.functional.scaled_dot_product_attention(
        q.reshape(B, Lq, heads, 32).transpose(1, 2).double(), k.reshape(B, Lk, heads, 32).transpose(1, 2).double(),
        v.reshape(B, Lk, heads, 32).transpose(1, 2).double(),
        attn_mask=None if mask is None else (mask[:, None, None, :] == 1)).transpose(1, 2).reshape(B, Lq, C)
    for mode, name in ((1, "f32"), (2, "bf16x6"), (3, "bf16x3")):
        ops.attention_mma(mode)
        out = ops.attention(q, k, v, heads, mask=mask)
        res[name] = (timeit(lambda: ops.attention(q, k, v, heads, mask=mask)),
                     float((out.double() - ref).abs().max() / ref.abs().max()))
    ops.attention_mma(-1)
    print(f"attn B{B} h{heads} {Lq}x{Lk}: " + " | ".join(
        f"{n} {res[n][0]:.1f} us {fl / res[n][0] * 1e-6:.1f} TF err {res[n][1]:.1e}" for n in res), flush=True)
