"""Latency of the §8(f) conditioners on the GPU at their real geometry (random-init weights, synthetic token ids / audio):
AudioMAE-token generator (GPT-2 base, KV-cached; 8 tokens = the text-to-audio configs, 512 = the speech config), FLAN-T5-large
encoder, CLAP text tower (RoBERTa-base), VITS phoneme encoder, CLAP audio tower (HTSAT-base) incl. the 16 -> 48 kHz resampler
= the candidate re-ranking of n_candidate_gen_per_text > 1.  Batch 8 (one GPU's prompts in bench.py)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from audioldm2_amd import ops  # noqa: E402,F401


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)


B = 8
g = torch.Generator().manual_seed(0)
torch.manual_seed(0)

from audioldm2_amd.t5 import FlanT5HiddenState  # noqa: E402
t5 = FlanT5HiddenState()
ids = torch.randint(3, 32000, (B, 32), generator=g)
ids[:, -1] = 1
am = torch.ones(B, 32, dtype=torch.long)
print(f"FLAN-T5-large encoder, {B} x 32 tokens: {timed(lambda: t5.encode_tokens(ids, am)):8.1f} ms", flush=True)
ids128 = torch.randint(3, 32000, (B, 128), generator=g)
am128 = torch.ones(B, 128, dtype=torch.long)
print(f"FLAN-T5-large encoder, {B} x 128 tokens: {timed(lambda: t5.encode_tokens(ids128, am128)):8.1f} ms", flush=True)
del t5
torch.cuda.empty_cache()

from audioldm2_amd.clap import CLAPAudioEmbeddingClassifierFreev2  # noqa: E402
clap = CLAPAudioEmbeddingClassifierFreev2(embed_mode="audio", unconditional_prob=0.0, sampling_rate=16000)
cid = torch.randint(3, 50000, (B, 512), generator=g)
cid[:, 0] = 0
cam = torch.zeros(B, 512, dtype=torch.long)
cam[:, :24] = 1
cid[cam == 0] = 1
clap.build_unconditional_emb({"input_ids": cid[:2], "attention_mask": cam[:2]})
print(f"CLAP text tower (RoBERTa-base), {B} x 512 positions: {timed(lambda: clap.encode_tokens(cid, cam)):8.1f} ms", flush=True)
wav = torch.randn(B, 163872, generator=g) * 0.1
print(f"CLAP audio tower (resample 16->48 kHz + HTSAT-base), {B} x 10.24 s: {timed(lambda: clap.encode_audio(wav)):8.1f} ms",
      flush=True)
tok = {"input_ids": cid, "attention_mask": cam}
print(f"CLAP re-ranking cos_similarity, {B} candidates: {timed(lambda: clap.cos_similarity(wav, tok)):8.1f} ms", flush=True)
del clap
torch.cuda.empty_cache()

from audioldm2_amd.phoneme import PhonemeEncoder  # noqa: E402
ph = PhonemeEncoder(vocabs_size=183, pad_length=310, pad_token_id=0)
pid = torch.randint(1, 183, (B, 310), generator=g)
pid[:, 200:] = 0
print(f"VITS phoneme encoder, {B} x 310 phonemes: {timed(lambda: ph(pid)):8.1f} ms", flush=True)
del ph

from audioldm2_amd.seqgen import Sequence2AudioMAE  # noqa: E402
for steps, keys, dims, T in ((8, ["film_clap_cond1", "crossattn_flan_t5"], [512, 1024], 32),
                             (512, ["film_clap_cond1", "crossattn_vits_phoneme"], [512, 192], 310)):
    m = Sequence2AudioMAE(sequence_gen_length=steps, sequence_input_key=keys, sequence_input_embed_dim=dims).cuda()
    cond = {keys[0]: torch.randn(B, 1, dims[0], generator=g).cuda(),
            keys[1]: [torch.randn(B, T, dims[1], generator=g).cuda(), torch.ones(B, T).cuda()]}
    ms = timed(lambda: m.generate(None, cond_dict=cond), reps=2)
    print(f"AudioMAE-token generator (GPT-2 base, KV-cached), {B} x {steps} tokens after a {T + 5}-position prompt: {ms:8.1f} ms "
          f"({ms / steps:.2f} ms per token)", flush=True)
    del m
