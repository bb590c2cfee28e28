"""Wall-clock breakdown of one text_to_audio-shaped job (audioldm2-full, B = 8, 200 steps) on the GPU:
conditioning, noise feed set-up, DDIM loop, VAE decode, vocoder, D2H.  Usage: python tools/job_breakdown.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm2_amd import ddim as D  # noqa: E402
from audioldm2_amd.pipeline import build_model, make_batch_for_text_to_audio, seed_everything  # noqa: E402

marks = []


def mark(name):
    torch.cuda.synchronize()
    marks.append((name, time.perf_counter()))


def main():
    torch.manual_seed(1234)
    ld = build_model(model_name="audioldm2-full").cuda()
    ld.scale_factor.fill_(0.75)
    batch = make_batch_for_text_to_audio("synthetic prompt", batchsize=8)
    seed_everything(42)
    ld.latent_t_size = 256
    for _ in range(2):  # warm: weights packed, graph cached
        ld.generate_batch(batch, unconditional_guidance_scale=3.5, ddim_steps=20, n_gen=1, duration=10)
    # instrument
    orig_sample_log, orig_dec, orig_voc = ld.sample_log, ld.decode_first_stage_cl, ld.mel_spectrogram_to_waveform
    orig_feed_init = D._NoiseFeed.__init__

    def sample_log(*a, **k):
        mark("cond+uncond+shard done")
        r = orig_sample_log(*a, **k)
        mark("sample_log (DDIM loop)")
        return r

    def dec(z):
        r = orig_dec(z)
        mark("VAE decode")
        return r

    def voc(*a, **k):
        r = orig_voc(*a, **k)
        mark("vocoder + D2H")
        return r

    def feed_init(self, *a, **k):
        mark("ddim setup before noise feed")
        orig_feed_init(self, *a, **k)
        mark("noise feed buffers")
    ld.sample_log, ld.decode_first_stage_cl, ld.mel_spectrogram_to_waveform = sample_log, dec, voc
    D._NoiseFeed.__init__ = feed_init
    for steps in (200,):
        marks.clear()
        mark("start")
        ld.generate_batch(batch, unconditional_guidance_scale=3.5, ddim_steps=steps, n_gen=1, duration=10)
        mark("end")
        t0 = marks[0][1]
        prev = t0
        print(f"--- {steps} DDIM steps, B=8")
        for name, t in marks[1:]:
            print(f"  {name:34s} +{(t - prev)*1e3:8.1f} ms   (t={(t - t0)*1e3:8.1f})")
            prev = t


if __name__ == "__main__":
    main()
