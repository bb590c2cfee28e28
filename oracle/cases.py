"""Seeded test cases shared by oracle/make_golden.py (which runs the REAL reference on them) and the
tests (which run the oracle restatement and the HIP product on them).  TEST INFRA ONLY.

Every input is regenerated from a seed, so fixtures only have to store reference OUTPUTS.
"""
from __future__ import annotations

import copy

import math

import torch

# --- UNet ----------------------------------------------------------------------------------------
UNET_TINY = {"image_size": 64, "context_dim": [768, 1024], "in_channels": 8, "out_channels": 8,
             "model_channels": 128, "attention_resolutions": [2, 1], "num_res_blocks": 1,
             "channel_mult": [1, 2], "num_head_channels": 32, "use_spatial_transformer": True,
             "transformer_depth": 1}
UNET_FULL = {"image_size": 64, "context_dim": [768, 1024], "in_channels": 8, "out_channels": 8,
             "model_channels": 128, "attention_resolutions": [8, 4, 2], "num_res_blocks": 2,
             "channel_mult": [1, 2, 3, 5], "num_head_channels": 32, "use_spatial_transformer": True,
             "transformer_depth": 1}
# -large-: third (context-free) transformer per location + depth 2 (utils.py:118-120), shrunk
UNET_LARGE_TINY = dict(UNET_TINY, context_dim=[768, 1024, None], transformer_depth=2)
# 48k: FiLM conditioning, no cross-attention context (utils.py:526-539), shrunk
UNET_FILM_TINY = {"image_size": 64, "extra_film_condition_dim": 512, "context_dim": [None], "in_channels": 16,
                  "out_channels": 16, "model_channels": 128, "attention_resolutions": [2, 1],
                  "num_res_blocks": 1, "channel_mult": [1, 2], "num_head_channels": 32,
                  "use_spatial_transformer": True, "transformer_depth": 1}


UNET_48K = {"image_size": 64, "extra_film_condition_dim": 512, "context_dim": [None], "in_channels": 16,
            "out_channels": 16, "model_channels": 128, "attention_resolutions": [8, 4, 2], "num_res_blocks": 2,
            "channel_mult": [1, 2, 3, 5], "num_head_channels": 32, "use_spatial_transformer": True,
            "transformer_depth": 1}  # utils.py:413-561 (audioldm_48k)


def unet_inputs(cfg: dict, B: int, H: int, W: int, t5_len: int = 12, seed: int = 0):
    """x, t, context_list, mask_list, y for a UNet config."""
    g = torch.Generator().manual_seed(1234 + seed)
    x = torch.randn(B, cfg["in_channels"], H, W, generator=g)
    t = torch.tensor([801, 6, 996, 301][:B] if B <= 4 else [(37 * i) % 1000 + 1 for i in range(B)])
    ctxs, masks = [], []
    for cd in cfg["context_dim"]:
        if cd is None:
            continue
        L = 8 if cd == 768 else t5_len
        ctxs.append(torch.randn(B, L, cd, generator=g))
        m = torch.ones(B, L)
        if cd != 768 and B > 1:
            m[1::2, -max(1, L // 3):] = 0
        masks.append(m)
    y = None
    if cfg.get("extra_film_condition_dim") is not None:
        y = torch.randn(B, cfg["extra_film_condition_dim"], generator=g)
        y = y / y.norm(dim=-1, keepdim=True)
    return x, t, ctxs, masks, y


# --- VAE / vocoder --------------------------------------------------------------------------------
DDCONFIG_16K = {"double_z": True, "mel_bins": 64, "z_channels": 8, "resolution": 256, "downsample_time": False,
                "in_channels": 1, "out_ch": 1, "ch": 128, "ch_mult": [1, 2, 4], "num_res_blocks": 2,
                "attn_resolutions": [], "dropout": 0}
DDCONFIG_48K = {"double_z": True, "mel_bins": 256, "z_channels": 16, "resolution": 256, "downsample_time": False,
                "in_channels": 1, "out_ch": 1, "ch": 128, "ch_mult": [1, 2, 4, 8], "num_res_blocks": 2,
                "attn_resolutions": [], "dropout": 0}
HIFIGAN_16K = dict(upsample_rates=[5, 4, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 4, 4],
                   upsample_initial_channel=1024, resblock_kernel_sizes=[3, 7, 11],
                   resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], num_mels=64, resblock="1")
HIFIGAN_48K = dict(upsample_rates=[6, 5, 4, 2, 2], upsample_kernel_sizes=[12, 10, 8, 4, 4],
                   upsample_initial_channel=1536, resblock_kernel_sizes=[3, 7, 11, 15],
                   resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5], [1, 3, 5]], num_mels=256,
                   resblock="1")


def latent_input(B, C, H, W, seed=0):
    return torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(77 + seed))


def mel_input(B, n_mel, T, seed=0):
    """log-mel-like values (roughly the range a trained VAE decodes to)."""
    g = torch.Generator().manual_seed(99 + seed)
    return torch.randn(B, n_mel, T, generator=g) * 2.0 - 4.0


def wave_input(B, T, seed=0):
    g = torch.Generator().manual_seed(55 + seed)
    x = 0.5 * (torch.rand(B, T, generator=g) * 2 - 1)
    tt = torch.arange(T, dtype=torch.float32) / 16000.0
    x[0] = 0.5 * torch.sin(2 * torch.pi * 440.0 * tt)  # 440 Hz known-answer row
    return x


# --- end-to-end ---------------------------------------------------------------------------------------
SCALE_FACTOR = 0.75
E2E_SEED = 42


def e2e_cond_config(device: str, t5_len: int = 32):
    from audioldm2_amd.pipeline import default_audioldm_config
    cond = copy.deepcopy(default_audioldm_config("audioldm2-full", t5_len)["model"]["params"]["cond_stage_config"])
    for k in cond:
        cond[k]["params"]["device"] = device
    return cond


def e2e_batch(B: int):
    """pipeline.py:112-121 batch layout (all-zero audio features, text tiled B times)."""
    text = ["a dog barking in the rain"] * B
    fbank = torch.zeros((B, 1024, 64))
    return {"text": text, "fname": [t.replace(" ", "_") for t in text], "waveform": torch.zeros((B, 160000)),
            "stft": torch.zeros((B, 1024, 512)), "log_mel_spec": fbank, "fbank": fbank,
            "ta_kaldi_fbank": torch.zeros((B, 1024, 128)), "phoneme_idx": torch.zeros((B, 310), dtype=torch.long)}


def e2e_masked_batch(B: int):
    """Inpainting / super-resolution input: the e2e batch with a real (seeded, log-mel-range) fbank
    [B, 1024, 64] in place of the all-zero one (pipeline.py:238-242)."""
    b = e2e_batch(B)
    fb = mel_input(B, 64, 1024, seed=11).permute(0, 2, 1).contiguous()
    b["log_mel_spec"] = fb
    b["fbank"] = fb
    return b


def e2e_batch_48k(B: int):
    """e2e_batch with the 256-bin all-zero log-mel of the 48 kHz model (utils.py:443-447)."""
    b = e2e_batch(B)
    b["log_mel_spec"] = torch.zeros((B, 1024, 256))
    b["fbank"] = b["log_mel_spec"]
    return b


# ---- §8(f) rank 1: AudioMAE-token sequence generator (sequence_input.py) -------------------------------------------
SEQGEN_FULL = dict(keys=["film_clap_cond1", "crossattn_flan_t5"], dims=[512, 1024], steps=8)        # utils.py:351-368
SEQGEN_SPEECH = dict(keys=["film_clap_cond1", "crossattn_vits_phoneme"], dims=[512, 192], steps=24)  # utils.py:124-143 (512 steps there)


def seqgen_cond(cfg: dict, B: int, T: int, seed: int = 5) -> dict:
    """Synthetic conditioning for the generator: a CLAP-like [B, 1, 512] vector (tensor-valued, all-ones mask) and a
    [B, T, D] sequence whose last quarter is padding (mask 0) for every odd sample."""
    g = torch.Generator().manual_seed(seed)
    film = torch.randn(B, 1, cfg["dims"][0], generator=g)
    seq = torch.randn(B, T, cfg["dims"][1], generator=g)
    mask = torch.ones(B, T)
    mask[1::2, T - T // 4:] = 0
    return {cfg["keys"][0]: film, cfg["keys"][1]: [seq, mask]}


# --- VITS phoneme encoder (second half; config 5) --------------------------------------------
PHONEME = {"vocabs_size": 183, "pad_token_id": 0, "pad_length": 310}   # utils.py:156-165


def phoneme_state_dict(shapes: dict, seed: int = 0) -> dict:
    """Deterministic weights for PhonemeEncoder; the relative-position tables and the positional embedding get O(0.1)
    values (their fan-in-scaled defaults would be ~1e-3 and a wrong relative-position shift would go unnoticed)."""
    import math
    from . import weights
    sd = weights.make_state_dict(shapes, seed=seed)
    for k, v in sd.items():
        if "emb_rel_" in k or k == "learnable_positional_embedding":
            fan = 1
            for s_ in v.shape[1:]:
                fan *= s_
            sd[k] = v * math.sqrt(fan) * 0.3
    return sd


def phoneme_input(seed: int = 7) -> torch.Tensor:
    """[4, 310] token ids: a full-length row, two padded rows (57 and 6 tokens: shorter than the +-4 window too) and a
    single-token row; pad id 0 at the end, as the reference's dataloader pads (commons.sequence_mask assumes it)."""
    g = torch.Generator().manual_seed(seed)
    T = PHONEME["pad_length"]
    idx = torch.randint(1, PHONEME["vocabs_size"], (4, T), generator=g)
    for b, n in enumerate((T, 57, 6, 1)):
        idx[b, n:] = PHONEME["pad_token_id"]
    return idx


# --- FLAN-T5 text conditioner ------------------------------------------------------------------
def t5_test_config() -> dict:
    """flan-t5-large's geometry (d_model 1024, 16 heads x 64, d_ff 2816, gated tanh-GELU, 32 buckets / distance 128) with 3
    of its 24 layers and a 512-entry vocabulary so that fixtures and deterministic weights stay small."""
    from .t5 import FLAN_T5_LARGE
    c = dict(FLAN_T5_LARGE)
    c.update(vocab_size=512, num_layers=3)
    return c


def t5_tokens(seed: int = 9):
    """Token ids as the reference's tokenizer call shapes them (padding=True: right padded with id 0, EOS id 1 last,
    modules.py:175-181): lengths 21, 7 and 1 (the empty prompt: EOS only) in a batch padded to 21."""
    g = torch.Generator().manual_seed(seed)
    T = 21
    ids = torch.randint(3, 512, (3, T), generator=g)
    mask = torch.zeros(3, T, dtype=torch.long)
    for b, n in enumerate((21, 7, 1)):
        ids[b, n - 1] = 1
        ids[b, n:] = 0
        mask[b, :n] = 1
    return ids, mask


def t5_state_dict(shapes: dict, seed: int = 0) -> dict:
    """Deterministic weights: T5's own initialisation scales (q: (d_model*d_kv)^-0.5 — it has no 1/sqrt(d) in the
    attention — so the softmax stays in a realistic regime), O(0.5) relative-position biases."""
    import math
    from . import weights
    sd = weights.make_state_dict(shapes, seed=seed)
    for k, v in sd.items():
        if k.endswith("SelfAttention.q.weight"):
            sd[k] = v * (64 ** -0.5) * 3.0
        elif "relative_attention_bias" in k:
            sd[k] = v * math.sqrt(v.shape[1]) * 0.8
        elif k in ("shared.weight", "encoder.embed_tokens.weight"):
            sd[k] = v * math.sqrt(v.shape[1])
    if "encoder.embed_tokens.weight" in sd:
        sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    return sd


# --- CLAP text tower (second half) --------------------------------------------------------------
def clap_text_test_config() -> dict:
    """roberta-base's geometry with 2 of its 12 layers, a 600-entry vocabulary and 66 positions (fixtures stay small)."""
    from .clap_text import ROBERTA_BASE
    c = dict(ROBERTA_BASE)
    c.update(vocab_size=600, num_hidden_layers=2, max_position_embeddings=66)
    return c


def clap_text_tokens(seed: int = 13):
    """[3, 64] ids at padding="max_length" (modules.py:737-745): <s> = 0 first, </s> = 2 last, pad = 1; lengths 64, 9, 2."""
    g = torch.Generator().manual_seed(seed)
    T = 64
    ids = torch.randint(3, 600, (3, T), generator=g)
    mask = torch.zeros(3, T, dtype=torch.long)
    for b, n in enumerate((64, 9, 2)):
        ids[b, 0] = 0
        ids[b, n - 1] = 2
        ids[b, n:] = 1
        mask[b, :n] = 1
    return ids, mask


# --- CLAP audio tower ----------------------------------------------------------------------------
def htsat_test_config() -> dict:
    """HTSAT-base's geometry (embed 128, heads 4/8/16/32 = head dim 32, window 8, 256x256 image) with depths (2, 2, 2, 2)
    instead of (2, 2, 12, 2): every kind of block (plain / shifted window, all four resolutions, patch merging) at 1/3 of
    the weights."""
    from .htsat import HTSAT_BASE
    c = dict(HTSAT_BASE)
    c["depths"] = (2, 2, 2, 2)
    return c


def htsat_state_dict(shapes: dict, seed: int = 0) -> dict:
    """Deterministic weights; relative-position tables O(0.5), BatchNorm statistics of a log-mel (mean ~ -30 dB, var ~ 100)."""
    import math
    from . import weights
    sd = weights.make_state_dict(shapes, seed=seed)
    for k, v in sd.items():
        if k.endswith("relative_position_bias_table"):
            sd[k] = v * math.sqrt(v.shape[1]) * 0.8
        elif k.endswith("bn0.running_mean"):
            sd[k] = -30.0 + 100.0 * v
        elif k.endswith("bn0.running_var"):
            sd[k] = 80.0 + 400.0 * v.abs()
    return sd


def clap_waveform(B: int = 2, seed: int = 21) -> torch.Tensor:
    """[B, 163872] at 16 kHz (the vocoder's output length): tones + noise bursts, |x| < 1."""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(163872) / 16000.0
    rows = []
    for b in range(B):
        f0 = 220.0 * (b + 1)
        x = 0.3 * torch.sin(2 * math.pi * f0 * t) + 0.2 * torch.sin(2 * math.pi * 3.1 * f0 * t + 1.0)
        x = x + 0.1 * torch.randn(t.shape, generator=g) * (torch.sin(2 * math.pi * 0.7 * t + b) > 0)
        rows.append(x)
    return torch.stack(rows).float()


# ---- conditioner drop-in end to end: stub tokenizers shared by the reference-side fixture generator and
# the HIP-side test (the Hub tokenizers are unreachable offline: parity starts at "the same token ids on both sides") --------
class StubT5Tokenizer:
    """Stands in for AutoTokenizer.from_pretrained("google/flan-t5-large") as FlanT5HiddenState calls it (encoders/modules.py:
    175-181: padding=True, truncation, return_tensors="pt"): deterministic ids from the characters of each prompt, EOS (1)
    last, right padded with 0 to the longest prompt; "" -> EOS alone.  Ids < 512 (cases.t5_test_config's vocabulary)."""

    def __call__(self, prompt, max_length=128, padding=True, truncation=True, return_tensors="pt"):
        import types
        prompt = [prompt] if isinstance(prompt, str) else list(prompt)
        rows = [[3 + (ord(ch) * 11 + i) % 500 for i, ch in enumerate(s)][: max_length - 1] + [1] for s in prompt]
        T = max(len(r) for r in rows)
        ids = torch.zeros(len(rows), T, dtype=torch.long)
        mask = torch.zeros(len(rows), T, dtype=torch.long)
        for b, r in enumerate(rows):
            ids[b, : len(r)] = torch.tensor(r)
            mask[b, : len(r)] = 1
        return types.SimpleNamespace(input_ids=ids, attention_mask=mask)


class StubRobertaTokenizer:
    """Stands in for RobertaTokenizer.from_pretrained("roberta-base") as CLAP calls it (encoders/modules.py:737-745:
    padding="max_length"): <s> = 0 first, </s> = 2 last, pad = 1, padded to 64; ids < 600 (cases.clap_text_test_config)."""

    def __call__(self, texts, padding=None, truncation=None, max_length=None, return_tensors=None):
        texts = [texts] if isinstance(texts, str) else list(texts)
        T = 64
        ids = torch.ones(len(texts), T, dtype=torch.long)
        mask = torch.zeros(len(texts), T, dtype=torch.long)
        for b, s in enumerate(texts):
            row = [0] + [3 + (ord(ch) * 7 + i) % 500 for i, ch in enumerate(s)][: T - 2] + [2]
            ids[b, : len(row)] = torch.tensor(row)
            mask[b, : len(row)] = 1
        return {"input_ids": ids, "attention_mask": mask}


E2E_COND_PROMPTS = ["a dog barking in the rain", "slow piano melody with soft strings"]


def e2e_cond_batch():
    b = e2e_batch(len(E2E_COND_PROMPTS))
    b["text"] = list(E2E_COND_PROMPTS)
    b["fname"] = [t.replace(" ", "_") for t in b["text"]]
    return b


def cond_state_dict(shapes: dict, seed: int = 0) -> dict:
    """Deterministic weights for the conditioner stack under `cond_stage_models.*`: name-keyed like everything else, with
    t5_state_dict's scalings for the tensors of the T5 encoders (q without 1/sqrt(d), relative-position biases, embeddings)."""
    import math
    from . import weights
    sd = weights.make_state_dict(shapes, seed=seed)
    for k, v in list(sd.items()):
        if k.endswith("SelfAttention.q.weight"):
            sd[k] = v * (64 ** -0.5) * 3.0
        elif "relative_attention_bias" in k:
            sd[k] = v * math.sqrt(v.shape[1]) * 0.8
        elif k.endswith(("shared.weight", "encoder.embed_tokens.weight")):
            sd[k] = v * math.sqrt(v.shape[1])
    for k in list(sd):
        if k.endswith("encoder.embed_tokens.weight"):
            sd[k] = sd[k[: -len("encoder.embed_tokens.weight")] + "shared.weight"]
    return sd
