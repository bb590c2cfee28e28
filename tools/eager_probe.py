"""Eager (no HIP graph) UNet pass, both matrix-core paths: wall time per pass and the host-side enqueue time
(python returns before the GPU is done) - tells a launch-bound pass from a GPU-bound one."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402
from audioldm2_amd.unet import UNetModel  # noqa: E402
from audioldm2_amd.pipeline import default_audioldm_config  # noqa: E402

cfg = default_audioldm_config("audioldm2-full")["model"]["params"]["unet_config"]["params"]
torch.manual_seed(0)
m = UNetModel(**cfg).cuda().eval()
g = torch.Generator().manual_seed(4)
x = torch.randn(16, 8, 256, 16, generator=g)
t = torch.tensor([(37 * i) % 1000 + 1 for i in range(16)])
ctxs = [torch.randn(16, 8, 768, generator=g), torch.randn(16, 32, 1024, generator=g)]
masks = [torch.ones(16, 8), torch.ones(16, 32)]
masks[1][1::2, -10:] = 0
xg, tg = x.cuda(), t.cuda()
cg, mg = [c.cuda() for c in ctxs], [k.cuda() for k in masks]
for mode in ("f32", "bf16x6", "f32", "bf16x6"):
    ops.set_mma(mode)
    with torch.no_grad():
        m(xg, tg, context_list=cg, context_attn_mask_list=mg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            m(xg, tg, context_list=cg, context_attn_mask_list=mg)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{mode}: enqueue {(t1 - t0) / 3 * 1e3:.1f} ms/pass, wall {(t2 - t0) / 3 * 1e3:.1f} ms/pass", flush=True)
