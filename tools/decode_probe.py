"""The speech model's 512-token AudioMAE-token generation (GPT-2 base, key/value cached, graph-replayed decode step) at batch 8
and at batch 4 (BASELINE config 5: 32 prompts over 8 GPUs), on the single-position decode kernels (csrc/decode.hip) and on the
general tile-GEMM path they replace (ALDM_SEQGEN_DECODE=general).  Usage: python tools/decode_probe.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from audioldm2_amd.seqgen import Sequence2AudioMAE  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 512
keys, dims, T = ["film_clap_cond1", "crossattn_vits_phoneme"], [512, 192], 310
g = torch.Generator().manual_seed(0)
torch.manual_seed(0)
m = Sequence2AudioMAE(sequence_gen_length=steps, sequence_input_key=keys, sequence_input_embed_dim=dims).cuda()
outs = {}
for B in [int(b) for b in os.environ.get("PROBE_B", "8,4").split(",")]:
    cond = {keys[0]: torch.randn(B, 1, dims[0], generator=g).cuda(),
            keys[1]: [torch.randn(B, T, dims[1], generator=g).cuda(), torch.ones(B, T).cuda()]}
    for mode in os.environ.get("PROBE_MODES", "general,fast,split,fast,split").split(","):
        os.environ["ALDM_SEQGEN_DECODE"] = mode
        m.generate(None, cond_dict=cond)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, _ = m.generate(None, cond_dict=cond)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        outs[(B, mode)] = out
        print(f"B={B} {steps} tokens after a {T + 5}-position prompt, decode={mode:8s}: {ms:8.1f} ms ({ms / steps:.3f} ms per token)",
              flush=True)
    if (B, "split") in outs and (B, "general") in outs:
        a, b = outs[(B, "split")].double(), outs[(B, "general")].double()
        print(f"B={B}: split vs general, max-norm rel diff over all {steps} tokens: {float((a - b).abs().max() / b.abs().max()):.2e}", flush=True)
    if (B, "fast") not in outs or (B, "general") not in outs:
        continue
    a, b = outs[(B, "fast")].double(), outs[(B, "general")].double()
    print(f"B={B}: fast vs general, max-norm rel diff over all {steps} tokens: {float((a - b).abs().max() / b.abs().max()):.2e}",
          flush=True)
