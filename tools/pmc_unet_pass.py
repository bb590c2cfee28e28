"""The command the HBM-traffic counter passes profile since round 5: UNet passes ONLY (two eager classifier-free-guidance passes of
audioldm2-full at batch 8 = what bench.py's roofline probe times), so that the per-instantiation averages of profiles/r05_pmc_traffic_*.json
are not mixed with the VAE / vocoder launches of the same instantiation (round 4's dominant kernel ran in the UNet only; round 5's —
igemm_dma_kernel<256, 128, ...> — also runs the decoder's 800-us convs).  Usage: ALDM_NO_GRAPH=1 rocprofv3 --pmc ... -- python tools/pmc_unet_pass.py [mode]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402
from audioldm2_amd.pipeline import build_model, make_batch_for_text_to_audio  # noqa: E402

if len(sys.argv) > 1:
    ops.set_mma(sys.argv[1])
B = 8
torch.manual_seed(1234)
ld = build_model(model_name="audioldm2-full").cuda()
ld.latent_t_size = 256
batch = make_batch_for_text_to_audio("synthetic prompt", batchsize=B)
cond = ld.get_learned_conditioning_dict(batch)
uncond = {k: ld.cond_stage_models[m["model_idx"]].get_unconditional_condition(B) for k, m in ld.cond_stage_model_metadata.items()}
x = torch.randn(B, ld.channels, ld.latent_t_size, ld.latent_f_size).cuda()
t2 = torch.full((2 * B,), 501.0).cuda()
for _ in range(2):
    ld.apply_model_cfg(x, t2, cond, uncond)
torch.cuda.synchronize()
print("2 UNet passes done")
