"""UNet DDIM step time (graph-replayed, bench.py's probe) of the library named by $ALDM_LIB_PATH — for same-box A/Bs.
Usage: [ALDM_LIB_PATH=...] python tools/step_probe.py [model] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from audioldm2_amd.pipeline import build_model, make_batch_for_text_to_audio  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "audioldm2-full"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = 8
torch.manual_seed(1234)
ld = build_model(model_name=model).cuda()
ld.latent_t_size = 256 if "48k" not in model else 128
batch = make_batch_for_text_to_audio("synthetic prompt", batchsize=B)
tag = os.path.basename(os.environ.get("ALDM_LIB_PATH", "libaldm_hip.so"))
for r in range(reps):
    print(f"{tag}: unet step {bench.unet_step_probe(ld, batch, B):.3f} ms", flush=True)
