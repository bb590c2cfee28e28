"""HiFi-GAN generator per batch of 8 (16 kHz: 1024 mel frames x 64 bins; 48 kHz: 1024 x 256) with the DMA-fed path taking stages down to
DMA_MIN_CHANNELS output channels (ALDM_HIFIGAN_DMA_MIN: 128 shipped, 64, 32), event-timed; outputs compared with the 128 setting.
Usage: python tools/hifigan_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import hifigan, ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else None
if mode:
    ops.set_mma(mode)
for name, bins in (("16k", 64), ("48k", 256)):
    torch.manual_seed(0)
    g = hifigan.get_vocoder(None, "cpu", bins).cuda()
    for p in g.parameters():          # variance-preserving gain, as the fixtures use (oracle/weights.VOCODER_GAIN)
        if p.dim() > 1:
            p.data.mul_(1.6)
    mel = torch.randn(8, 1024, bins).cuda()
    ref = None
    for dmin in (128, 64, 32):
        hifigan.Generator.DMA_MIN_CHANNELS = dmin
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = g.forward_cl(mel)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        if ref is None:
            ref = y.clone()
        d = float((y - ref).abs().max() / ref.abs().max())
        print(f"{name} {ops.MMA_MODE}: DMA stages with >= {dmin:3d} channels: {ms:8.2f} ms per 8 clips, max-norm diff vs 128: {d:.2e}, finite {bool(torch.isfinite(y).all())}", flush=True)
