"""Time one DMA-fed igemm launch (conv3x3 128->128 @256x16, 16 samples, forced 128x128 tile, 3-deep ring) with the library
named by $ALDM_LIB_PATH — ablation builds (ALDM_DMA_ABLATE, csrc/igemm_dma.h) drop pieces of the K loop; results of those
builds are wrong by construction, only the time means something."""
import math
import os
import sys

import torch

sys.path.insert(0, ".")
from audioldm2_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
tag = os.path.basename(os.environ.get("ALDM_LIB_PATH", "libaldm_hip.so"))
for name, B, H, W, C, N, k, tile in [("conv3x3 128->128", 16, 256, 16, 128, 128, 3, (128, 128, 3)),
                                     ("linear 16384x256->2048", 1, 1, 16384, 256, 2048, 1, (128, 128, 3)),
                                     ("conv3x3 128->128 256x128", 16, 256, 16, 128, 128, 3, (256, 128, 2))]:
    x = torch.randn(B, H, W, C, generator=g).cuda()
    w = torch.randn(N, C, k, k, generator=g) / math.sqrt(C * k * k)
    pw = ops.pack_conv(w, None)
    xs = ops.split_rows(x)
    ops.igemm_force(tile[0], tile[1], 1, 0, tile[2])
    fn = lambda: ops.conv(xs, pw, pad=(k // 2, k // 2))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ops.igemm_force(0, 0, 0)
    t = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{tag:28s} {name:26s} {t:7.1f} us  {2.0 * B * H * W * N * C * k * k / t * 1e-6:6.1f} TF", flush=True)
