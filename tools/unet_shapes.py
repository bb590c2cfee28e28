"""Per-shape table of every igemm launch of ONE UNet CFG pass (audioldm2-full, B prompts -> 2B samples),
timed with events on the launch stream.  Usage (GPU box): python tools/unet_shapes.py [B] [model]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ops  # noqa: E402
from audioldm2_amd.pipeline import build_model, make_batch_for_text_to_audio  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    model = sys.argv[2] if len(sys.argv) > 2 else "audioldm2-full"
    torch.manual_seed(1234)
    ld = build_model(model_name=model).cuda()
    batch = make_batch_for_text_to_audio("synthetic prompt", batchsize=B)
    cond = ld.get_learned_conditioning_dict(batch)
    uncond = {k: ld.cond_stage_models[m["model_idx"]].get_unconditional_condition(B)
              for k, m in ld.cond_stage_model_metadata.items()}
    x = torch.randn(B, ld.channels, ld.latent_t_size, ld.latent_f_size).cuda()
    t2 = torch.full((2 * B,), 501.0).cuda()
    for _ in range(2):
        ld.apply_model_cfg(x, t2, cond, uncond)
    torch.cuda.synchronize()
    reps = 3
    agg = {}
    for _ in range(reps):
        ops.PROFILE = []
        ld.apply_model_cfg(x, t2, cond, uncond)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        for what, bm, bn, fl, e0, e1, shape, _kn in prof:
            a = agg.setdefault((shape, bm, bn), [0, 0.0, 0.0])
            a[0] += 1
            a[1] += fl
            a[2] += e0.elapsed_time(e1) * 1e-3
    tot_t = sum(v[2] for v in agg.values()) / reps
    tot_f = sum(v[1] for v in agg.values()) / reps
    print(f"# {model} B={B} (2B={2*B} samples/pass): igemm {tot_t*1e3:.2f} ms/pass, {tot_f/tot_t/1e12:.1f} TFLOP/s")
    print(f"{'M':>7s} {'N':>5s} {'K':>6s} {'taps':>4s} {'C2':>5s} pre act bat  sp {'tile':>8s} {'n':>4s} {'us/call':>9s} {'ms/pass':>8s} {'pct':>5s} {'TF/s':>6s}")
    for (shape, bm, bn), (n, fl, sec) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        M, N, K, taps, C2, pre, act, bat, sp = shape
        n1 = n / reps
        print(f"{M:7d} {N:5d} {K:6d} {taps:4d} {C2:5d} {pre:3d} {act:3d} {bat:3d} {sp:3d} {bm:4d}x{bn:<3d} {n1:4.0f} "
              f"{sec/n*1e6:9.1f} {sec/reps*1e3:8.3f} {100*sec/reps/tot_t:5.1f} {fl/sec/1e12:6.1f}")


if __name__ == "__main__":
    main()
