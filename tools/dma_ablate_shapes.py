"""Where the time of the short-K DMA-fed GEMMs of the transformer blocks goes: a few geometries of the audioldm2-full step
(batch 16 incl. CFG) x a few tile configurations, timed with the library named by $ALDM_LIB_PATH.  Ablation builds
(ALDM_DMA_ABLATE, csrc/igemm_dma.h: 16 = no epilogue, 32 = one k-tile only, 4 = no MFMA, 3 = no DMA) give wrong results by
construction; only their times mean something."""
import math
import os
import sys

import torch

sys.path.insert(0, ".")
from audioldm2_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
tag = os.path.basename(os.environ.get("ALDM_LIB_PATH", "libaldm_hip.so")).replace("libaldm_", "").replace(".so", "")
SHAPES = [  # name, M, K, N, kind
    ("geglu 16384x256->2x1024", 16384, 256, 2048, "geglu"),
    ("qkv   16384x256->768", 16384, 256, 768, "plain"),
    ("proj  16384x256->256+res", 16384, 256, 256, "res"),
    ("ffout 16384x1024->256+res", 16384, 1024, 256, "res"),
    ("l3    1024x640->640+res", 1024, 640, 640, "res"),
    ("l2 geglu 4096x384->2x1536", 4096, 384, 3072, "geglu"),
]
TILES = [(0, 0, 0), (64, 128, 2), (128, 128, 2), (256, 128, 2), (64, 64, 3)]
for name, M, K, N, kind in SHAPES:
    x = torch.randn(M, K, generator=g).cuda()
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    res = torch.randn(M, N, generator=g).cuda() if kind == "res" else None
    pw = ops.pack_geglu(w, b) if kind == "geglu" else ops.pack_conv(w, b)
    xs = ops.split_rows(x.view(1, 1, M, K))
    row = []
    for tile in TILES:
        ops.igemm_force(tile[0], tile[1], 1 if tile[0] else 0, 0, tile[2])
        if kind == "geglu":
            fn = lambda: ops.linear_geglu(xs, pw, split_out="only")
        elif kind == "res":
            fn = lambda: ops.linear(xs, pw, res=res)
        else:
            fn = lambda: ops.linear(xs, pw)
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            # 20 launches per HIP-graph replay: the eager ctypes launch path (~17 us per call) would hide the short kernels
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
            row.append(f"{e0.elapsed_time(e1) * 1e3 / 100:6.1f}")
        except Exception as ex:  # a forced tile the shape does not admit
            row.append("   n/a")
        ops.igemm_force(0, 0, 0)
    print(f"{tag:10s} {name:28s} " + "  ".join(f"{t[0]}x{t[1]}s{t[2]}:{r}" for t, r in zip(TILES, row)), flush=True)
