"""LayerNorm (+ split image) launch times on the UNet's token shapes, 20 launches per HIP-graph replay."""
import sys

import torch

sys.path.insert(0, ".")
from audioldm2_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
for M, C in ((16384, 256), (4096, 384), (1024, 640), (65536, 128)):
    x = torch.randn(M, C, generator=g).cuda()
    ga, be = torch.ones(C).cuda(), torch.zeros(C).cuda()
    for so in ("only", None):
        fn = lambda: ops.layernorm(x, ga, be, split_out=so)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            fn()
            with torch.cuda.graph(gr, stream=side):
                for _ in range(20):
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
        t = e0.elapsed_time(e1) * 1e3 / 100
        print(f"layernorm M={M} C={C} split_out={so}: {t:6.1f} us  {8.0 * M * C / t * 1e-6:5.2f} TB/s", flush=True)
