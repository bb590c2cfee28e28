"""A/B of the 8-wavefront tile variants on the UNet DDIM step (graph-replayed, bench.py's step probe).
Usage: python tools/wave8_ab.py [model] [masks...]   (masks: aldm_igemm_wave8_mask bit masks, default 0 1 3 5 7)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from audioldm2_amd import ops  # noqa: E402
from audioldm2_amd.pipeline import build_model, make_batch_for_text_to_audio  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "audioldm2-full"
masks = [int(a) for a in sys.argv[2:]] or [0, 1, 3, 5, 7]
B = 8
torch.manual_seed(1234)
ld = build_model(model_name=model).cuda()
ld.latent_t_size = 256 if "48k" not in model else 128
batch = make_batch_for_text_to_audio("synthetic prompt", batchsize=B)
unet = ld.model.diffusion_model
for rep in range(2):
    for m in masks:
        ops.igemm_wave8(m)
        unet._graph_cache.clear()
        ms = bench.unet_step_probe(ld, batch, B)
        print(f"model {model} rep {rep} wave8 mask {m}: unet step {ms:.3f} ms", flush=True)
ops.igemm_wave8(-1)
