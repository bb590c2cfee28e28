"""oracle/htsat.py — CPU restatement of the CLAP AUDIO tower (re-ranking of n_candidate_gen_per_text > 1).
TEST INFRASTRUCTURE ONLY.

Reference: `CLAPAudioEmbeddingClassifierFreev2.forward` in "audio" mode (encoders/modules.py:689-716) -> resample to 48 kHz ->
`get_audio_features` (clap/training/data.py:421-450: with fusion disabled only `waveform[..., :480000]` is used) ->
`CLAP.get_audio_embedding` (clap/open_clip/model.py:749-775) -> `HTSAT_Swin_Transformer.forward` (clap/open_clip/htsat.py:1111-1127
non-fusion branch, `reshape_wav2img` :1064-1090, `forward_features` :1010-1053 — only `"embedding"` is consumed) ->
`audio_projection` (model.py:563-567) -> F.normalize; and `cos_similarity` (encoders/modules.py:639-653).
Restated from code that is NOT in the reference tree (parity UNPINNED for these, named with the pins of requirements.txt):
  * torchaudio.functional.resample (torchaudio==0.13.1): windowed-sinc polyphase FIR, lowpass_filter_width 6, rolloff 0.99,
    Hann window (`_get_sinc_resample_kernel` / `_apply_sinc_resample_kernel`);
  * torchlibrosa==0.0.9 `Spectrogram` (DFT-basis conv1d, periodic Hann, centre reflect pad, power 2) and `LogmelFilterBank`
    (librosa mel basis — Slaney scale and norm — then 10*log10(clamp(x, amin)), ref 1, top_db None): htsat.py:889-909.
Everything else (bn0, bicubic reshape, Swin blocks, pooling) is pinned by a fixture generated from the REAL
`HTSAT_Swin_Transformer` with those two extractors injected (tests/golden/htsat_*.npz, oracle/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import stft as ostft

AUDIO_CFG = dict(sample_rate=48000, window_size=1024, hop_size=480, mel_bins=64, fmin=50, fmax=14000, clip_samples=480000)
HTSAT_BASE = dict(embed_dim=128, depths=(2, 2, 12, 2), num_heads=(4, 8, 16, 32), window_size=8, spec_size=256, patch=4,
                  mlp_ratio=4.0)
BN_EPS, LN_EPS = 1e-5, 1e-5


# ---- torchaudio.functional.resample (sinc_interp_hann) -----------------------------------------------------------------
def sinc_resample_kernel(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    g = math.gcd(orig, new)
    orig, new = orig // g, new // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base / orig
    k = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * scale
    return k.to(torch.float32), width, orig, new           # [new, 1, 2*width + orig]


def resample(x: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """x [B, T] -> [B, ceil(new * T / orig)]"""
    k, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
    T = x.shape[-1]
    y = F.conv1d(F.pad(x[:, None], (width, width + orig)), k, stride=orig)      # [B, new, n]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    return y[:, : int(math.ceil(new * T / orig))]


# ---- torchlibrosa front end ------------------------------------------------------------------------------------------
def power_spectrogram(x: torch.Tensor, n_fft: int, hop: int) -> torch.Tensor:
    """x [B, T] -> [B, frames, n_fft//2 + 1] (torchlibrosa Spectrogram, power 2)."""
    basis = torch.from_numpy(ostft.stft_forward_basis(n_fft, n_fft))[:, None, :]
    xp = F.pad(x[:, None, None, :], (n_fft // 2, n_fft // 2, 0, 0), mode="reflect")[:, 0]
    ft = F.conv1d(xp, basis, stride=hop)
    c = n_fft // 2 + 1
    return (ft[:, :c] ** 2 + ft[:, c:] ** 2).transpose(1, 2)


def logmel(power: torch.Tensor, cfg: dict, amin: float = 1e-10) -> torch.Tensor:
    """[B, frames, F] -> [B, frames, mel_bins]: power @ melW then 10*log10(clamp(., amin)) (ref = 1, top_db None)."""
    melW = torch.from_numpy(ostft.mel_filterbank(cfg["sample_rate"], cfg["window_size"], cfg["mel_bins"], cfg["fmin"],
                                                 cfg["fmax"])).t()
    return 10.0 * torch.log10(torch.clamp(power @ melW, min=amin))


# ---- HTSAT (htsat.py) -------------------------------------------------------------------------------------------------
def relative_position_index(ws: int) -> torch.Tensor:
    """htsat.py:371-386"""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_attn_mask(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    """htsat.py:527-553: [nW, ws*ws, ws*ws] of 0 / -100."""
    img = torch.zeros((1, H, W, 1))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, h, w, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def window_partition(x, ws):
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(win, ws, H, W):
    B = int(win.shape[0] / (H * W / ws / ws))
    return win.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def swin_block(sd, p: str, x: torch.Tensor, H: int, W: int, heads: int, ws: int, shift: int) -> torch.Tensor:
    """SwinTransformerBlock.forward (htsat.py:556-596) + WindowAttention.forward (:396-441)."""
    if min(H, W) <= ws:
        shift, ws = 0, min(H, W)
    B, L, C = x.shape
    ln = lambda t, n: F.layer_norm(t, (C,), sd[p + n + ".weight"], sd[p + n + ".bias"], LN_EPS)
    h = ln(x, "norm1").view(B, H, W, C)
    if shift > 0:
        h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
    win = window_partition(h, ws).view(-1, ws * ws, C)
    B_, N, _ = win.shape
    qkv = F.linear(win, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = sd[p + "attn.relative_position_bias_table"][relative_position_index(ws).view(-1)].view(N, N, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shift > 0:
        m = shift_attn_mask(H, W, ws, shift)
        nW = m.shape[0]
        attn = (attn.view(B_ // nW, nW, heads, N, N) + m.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    a = (F.softmax(attn, dim=-1) @ v).transpose(1, 2).reshape(B_, N, C)
    a = F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    h = window_reverse(a.view(-1, ws, ws, C), ws, H, W)
    if shift > 0:
        h = torch.roll(h, shifts=(shift, shift), dims=(1, 2))
    x = x + h.view(B, L, C)
    m = F.linear(F.gelu(F.linear(ln(x, "norm2"), sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])),
                 sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + m


def patch_merging(sd, p: str, x: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """PatchMerging.forward (htsat.py:653-676)"""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
    x = F.layer_norm(x, (4 * C,), sd[p + "norm.weight"], sd[p + "norm.bias"], LN_EPS)
    return F.linear(x, sd[p + "reduction.weight"])


def reshape_wav2img(x: torch.Tensor, spec_size: int, freq_ratio: int) -> torch.Tensor:
    """htsat.py:1064-1090: [B, 1, T, F] -> [B, 1, spec_size, spec_size] (time folded into `freq_ratio` row bands)."""
    target_T, target_F = spec_size * freq_ratio, spec_size // freq_ratio
    if x.shape[2] < target_T:
        x = F.interpolate(x, (target_T, x.shape[3]), mode="bicubic", align_corners=True)
    if x.shape[3] < target_F:
        x = F.interpolate(x, (x.shape[2], target_F), mode="bicubic", align_corners=True)
    x = x.permute(0, 1, 3, 2).contiguous()
    x = x.reshape(x.shape[0], x.shape[1], x.shape[2], freq_ratio, x.shape[3] // freq_ratio)
    x = x.permute(0, 1, 3, 2, 4).contiguous()
    return x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3], x.shape[4])


def htsat_embedding(sd: Dict[str, torch.Tensor], waveform48: torch.Tensor, hcfg: dict = None, acfg: dict = None,
                    prefix: str = "audio_branch.") -> torch.Tensor:
    """HTSAT_Swin_Transformer.forward(...)["embedding"] (eval, no fusion): waveform [B, T] at 48 kHz -> [B, 8 * embed_dim]."""
    hcfg, acfg = hcfg or HTSAT_BASE, acfg or AUDIO_CFG
    x = logmel(power_spectrogram(waveform48, acfg["window_size"], acfg["hop_size"]), acfg)[:, None]   # [B, 1, T, mel]
    bn = lambda n: sd[prefix + "bn0." + n]
    x = (x - bn("running_mean")) / torch.sqrt(bn("running_var") + BN_EPS) * bn("weight") + bn("bias")   # :1118-1120
    x = reshape_wav2img(x, hcfg["spec_size"], hcfg["spec_size"] // acfg["mel_bins"])
    x = F.conv2d(x, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"], stride=hcfg["patch"])
    x = x.flatten(2).transpose(1, 2)
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), sd[prefix + "patch_embed.norm.weight"], sd[prefix + "patch_embed.norm.bias"], LN_EPS)
    H = W = hcfg["spec_size"] // hcfg["patch"]
    for i, depth in enumerate(hcfg["depths"]):
        for j in range(depth):
            x = swin_block(sd, f"{prefix}layers.{i}.blocks.{j}.", x, H, W, hcfg["num_heads"][i], hcfg["window_size"],
                           0 if j % 2 == 0 else hcfg["window_size"] // 2)
        if i < len(hcfg["depths"]) - 1:
            x = patch_merging(sd, f"{prefix}layers.{i}.downsample.", x, H, W)
            H, W = H // 2, W // 2
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], LN_EPS)
    return x.mean(dim=1)      # :1034-1035: avgpool over every (frequency, time) position of the rearranged map


def audio_embedding(sd: Dict[str, torch.Tensor], waveform: torch.Tensor, sampling_rate: int, hcfg: dict = None,
                    acfg: dict = None) -> torch.Tensor:
    """encoders/modules.py:689-716 + model.py:749-775: waveform [B, T] at `sampling_rate` -> L2-normalised [B, 512]."""
    acfg = acfg or AUDIO_CFG
    if sampling_rate != 48000:
        waveform = resample(waveform, sampling_rate, 48000)
    e = htsat_embedding(sd, waveform[:, : acfg["clip_samples"]], hcfg, acfg)
    e = F.linear(e, sd["audio_projection.0.weight"], sd["audio_projection.0.bias"])
    e = F.linear(torch.relu(e), sd["audio_projection.2.weight"], sd["audio_projection.2.bias"])
    return F.normalize(e, dim=-1)


def cos_similarity(audio_emb: torch.Tensor, text_emb: torch.Tensor) -> torch.Tensor:
    """encoders/modules.py:651: both [B, 1, 512] in the reference; here [B, 512] -> [B]."""
    return F.cosine_similarity(audio_emb, text_emb, dim=-1)
