"""fp32-MFMA vs bf16-split attention (aldm_attention_mma 1 / 2) on the UNet's self-attention shapes, same process."""
import sys

import torch

sys.path.insert(0, ".")
from audioldm2_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


g = torch.Generator().manual_seed(0)
for B, heads, Lq, Lk in [(16, 8, 1024, 1024), (16, 12, 256, 256), (16, 20, 64, 64), (16, 8, 1024, 32)]:
    qkv = torch.randn(B, Lq, 3 * heads * 32, generator=g).cuda()
    kv = torch.randn(B, Lk, 2 * heads * 32, generator=g).cuda()
    C = heads * 32
    q, k, v = (qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]) if Lq == Lk else (qkv[:, :, :C], kv[:, :, :C], kv[:, :, C:])
    fl = 4.0 * B * heads * Lq * Lk * 32
    res = {}
    for mode, name in ((1, "f32"), (2, "bf16x6")):
        ops.attention_mma(mode)
        res[name] = (timeit(lambda: ops.attention(q, k, v, heads)), ops.attention(q, k, v, heads))
    ops.attention_mma(-1)
    err = float((res["f32"][1] - res["bf16x6"][1]).abs().max() / res["f32"][1].abs().max())
    print(f"attn B{B} h{heads} {Lq}x{Lk}: f32 {res['f32'][0]:.1f} us {fl / res['f32'][0] * 1e-6:.1f} TF | bf16x6 "
          f"{res['bf16x6'][0]:.1f} us {fl / res['bf16x6'][0] * 1e-6:.1f} TF | diff {err:.1e}", flush=True)
