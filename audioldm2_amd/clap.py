"""CLAP on the MI355X — text tower (second half) and audio tower (rank 4: candidate re-ranking) —
behind the reference's conditioner class.

`CLAPAudioEmbeddingClassifierFreev2` mirrors `audioldm2.latent_diffusion.modules.encoders.modules.CLAPAudioEmbeddingClassifierFreev2`
(encoders/modules.py:546-745): same constructor keywords, `forward(batch)` in `embed_mode="text"` (the mode every AudioLDM2
config uses for the FiLM / sequence conditioning, utils.py:146-155), `get_unconditional_condition`, and the
`unconditional_prob` draw of modules.py:731-733 (a `torch.rand(1)` per batch element from the host generator — RNG contract R2).
`self.model` holds the text side of `clap.open_clip.model.CLAP` under its state-dict keys (`text_branch.*` = transformers
RobertaModel, `text_projection.{0,2}.*`), so `cond_stage_models.*.model.text_*` / `clap.model.text_*` checkpoint entries load
load; `audio_branch.*` (HTSAT_Swin_Transformer, clap/open_clip/htsat.py) and `audio_projection.{0,2}.*` sit next to them.

embed_mode="audio" (encoders/modules.py:689-716) and `cos_similarity` (:639-653: CLAP re-ranking of
n_candidate_gen_per_text > 1, ddpm.py:1554-1568): waveform -> torchaudio-style sinc resampling to 48 kHz -> [:480000] ->
HTSAT "embedding" (htsat.py:1111-1127 non-fusion branch: torchlibrosa power spectrogram + log-mel, bn0, reshape_wav2img,
4x4 patch embedding, 4 stages of (shifted-)window attention blocks with patch merging, final LayerNorm, mean over tokens)
-> audio_projection -> F.normalize.  Fusion (`enable_fusion`) is off in every AudioLDM2 use and not built.

The tokenizer is the reference's `RobertaTokenizer.from_pretrained("roberta-base")`; where the Hub is out of reach the module
still builds (roberta-base's published geometry) and `encode_tokens(input_ids, attention_mask)` is the entry point.
Compute: embedding gathers by torch, everything else through the C ABI (aldm_layernorm, aldm_igemm with fused bias / erf-GELU /
tanh / ReLU / residual epilogues, Q K^T and P V as batched GEMMs around aldm_softmax_rows_masked, aldm_row_l2norm +
aldm_rowscale_add for F.normalize).  Audio tower: aldm_resample_sinc, the STFT as the frames GEMM of the mel front end,
aldm_power_spec, mel projection with the log-clamp epilogue, aldm_col_affine (bn0 and the dB factor), aldm_bicubic_patchify,
window partition / cyclic shift / patch merging as row gathers (torch index_select: pure data movement), attention scores as
batched GEMMs around aldm_softmax_rows_bias (relative-position bias + shift mask), aldm_token_mean, aldm_row_cosine.
These GEMMs take fp32 operands, i.e. the engine's exact bf16x6 products: the log-mel of quiet bins needs fp32-grade sums.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .stft import STFT, mel_filterbank

ROBERTA_BASE = dict(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                    max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1)
JOINT_DIM = 512
# Where the packed weights and every activation live.  The product has one answer; tests/test_host_logic.py points this at
# the CPU together with torch stand-ins for `ops` to check the host logic (window indices, bias assembly, gathers) on a
# GPU-less box — the real ops raise on CPU tensors.
_DEV = torch.device("cuda")
# clap/open_clip/model_configs/HTSAT-base.json + htsat.py:1224-1236 ("base")
AUDIO_CFG = dict(sample_rate=48000, window_size=1024, hop_size=480, mel_bins=64, fmin=50, fmax=14000, clip_samples=480000,
                 class_num=527)
HTSAT_BASE = dict(embed_dim=128, depths=(2, 2, 12, 2), num_heads=(4, 8, 16, 32), window_size=8, spec_size=256, patch=4,
                  mlp_ratio=4.0)


class _Emb(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        h = cfg["hidden_size"]
        self.word_embeddings = nn.Embedding(cfg["vocab_size"], h)
        self.position_embeddings = nn.Embedding(cfg["max_position_embeddings"], h)
        self.token_type_embeddings = nn.Embedding(cfg["type_vocab_size"], h)
        self.LayerNorm = nn.LayerNorm(h)


class _SelfAttn(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(h, h), nn.Linear(h, h), nn.Linear(h, h)


class _Out(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.dense = nn.Linear(i, o)
        self.LayerNorm = nn.LayerNorm(o)


class _Attn(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.self = _SelfAttn(h)
        self.output = _Out(h, h)


class _Inter(nn.Module):
    def __init__(self, h, i):
        super().__init__()
        self.dense = nn.Linear(h, i)


class _Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        h, i = cfg["hidden_size"], cfg["intermediate_size"]
        self.attention = _Attn(h)
        self.intermediate = _Inter(h, i)
        self.output = _Out(i, h)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(cfg) for _ in range(cfg["num_hidden_layers"])])


class _Pooler(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.dense = nn.Linear(h, h)


class _Roberta(nn.Module):
    """Parameter holder with transformers.RobertaModel's names."""

    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Emb(cfg)
        self.encoder = _Encoder(cfg)
        self.pooler = _Pooler(cfg["hidden_size"])


# ---- HTSAT audio tower: parameter holders under the reference's names (clap/open_clip/htsat.py) -----------------------
def _relative_position_index(ws: int) -> torch.Tensor:
    """htsat.py:371-386"""
    ar = torch.arange(ws)
    coords = torch.stack(torch.meshgrid(ar, ar, indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _window_index(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    """Token order of `window_partition(roll(x, -shift))` (htsat.py:563-575): out[win * ws*ws + i*ws + j] = source token."""
    r = (torch.arange(H).view(H // ws, 1, ws, 1) + shift) % H            # [wh, 1, i, 1]
    c = (torch.arange(W).view(1, W // ws, 1, ws) + shift) % W            # [1, ww, 1, j]
    return (r * W + c).reshape(-1)


def _shift_attn_mask(H: int, W: int, ws: int, shift: int) -> torch.Tensor:
    """htsat.py:527-553: [nW, ws*ws, ws*ws] of 0 / -100 between tokens that the cyclic shift brought together."""
    img = torch.zeros(H, W)
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[h, w] = cnt
            cnt += 1
    mw = img.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(am != 0, torch.full_like(am, -100.0), torch.zeros_like(am))


class _WindowAttention(nn.Module):
    def __init__(self, dim, ws, heads):
        super().__init__()
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.register_buffer("relative_position_index", _relative_position_index(ws))
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)


class _SwinBlock(nn.Module):
    def __init__(self, dim, res, heads, ws, shift, mlp_ratio):
        super().__init__()
        if min(res) <= ws:                      # htsat.py:492-495
            shift, ws = 0, min(res)
        self.res, self.heads, self.ws, self.shift = res, heads, ws, shift
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _WindowAttention(dim, ws, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        if shift > 0:
            self.register_buffer("attn_mask", _shift_attn_mask(res[0], res[1], ws, shift))
        else:
            self.attn_mask = None


class _PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)


class _BasicLayer(nn.Module):
    def __init__(self, dim, res, depth, heads, ws, mlp_ratio, downsample):
        super().__init__()
        self.blocks = nn.ModuleList([_SwinBlock(dim, res, heads, ws, 0 if i % 2 == 0 else ws // 2, mlp_ratio)
                                     for i in range(depth)])
        self.downsample = _PatchMerging(dim) if downsample else None


class _PatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.proj = nn.Conv2d(1, dim, kernel_size=patch, stride=patch)
        self.norm = nn.LayerNorm(dim)


class _Holder(nn.Module):
    pass


class HTSAT(nn.Module):
    """State-dict twin of `HTSAT_Swin_Transformer(config = HTSAT-base, enable_fusion=False)` (htsat.py:777-1010), including the
    two torchlibrosa extractors' frozen tensors (`spectrogram_extractor.stft.conv_{real,imag}.weight`,
    `logmel_extractor.melW` — names as torchlibrosa 0.0.9 registers them; that package is not vendored, so they are
    restated: DFT basis x periodic Hann, librosa Slaney mel basis) and the unused classifier head (`tscam_conv`, `head`)."""

    def __init__(self, hcfg: dict = None, acfg: dict = None):
        super().__init__()
        self.hcfg, self.acfg = dict(hcfg or HTSAT_BASE), dict(acfg or AUDIO_CFG)
        h, a = self.hcfg, self.acfg
        n_fft, F = a["window_size"], a["window_size"] // 2 + 1
        basis = STFT(n_fft, a["hop_size"], n_fft, "hann").forward_basis           # [2F, 1, n_fft]
        self.spectrogram_extractor = _Holder()
        self.spectrogram_extractor.stft = _Holder()
        cr, ci = nn.Conv1d(1, F, n_fft, stride=a["hop_size"], bias=False), nn.Conv1d(1, F, n_fft, stride=a["hop_size"], bias=False)
        cr.weight.data.copy_(basis[:F])
        ci.weight.data.copy_(basis[F:])
        self.spectrogram_extractor.stft.conv_real, self.spectrogram_extractor.stft.conv_imag = cr, ci
        self.logmel_extractor = _Holder()
        self.logmel_extractor.melW = nn.Parameter(torch.from_numpy(np.ascontiguousarray(
            mel_filterbank(a["sample_rate"], n_fft, a["mel_bins"], a["fmin"], a["fmax"]).T)), requires_grad=False)
        self.bn0 = nn.BatchNorm2d(a["mel_bins"])
        dim, nl = h["embed_dim"], len(h["depths"])
        self.patch_embed = _PatchEmbed(h["patch"], dim)
        g = h["spec_size"] // h["patch"]
        self.layers = nn.ModuleList([
            _BasicLayer(dim * 2 ** i, (g // 2 ** i, g // 2 ** i), h["depths"][i], h["num_heads"][i], h["window_size"],
                        h["mlp_ratio"], i < nl - 1) for i in range(nl)])
        self.num_features = dim * 2 ** (nl - 1)
        self.norm = nn.LayerNorm(self.num_features)
        sf = h["spec_size"] // 2 ** (nl - 1) // h["patch"] // (h["spec_size"] // a["mel_bins"])
        self.tscam_conv = nn.Conv2d(self.num_features, a["class_num"], kernel_size=(sf, 3), padding=(0, 1))
        self.head = nn.Linear(a["class_num"], a["class_num"])
        self._pk = None

    def invalidate_packed(self):
        self._pk = None

    def _prepare(self):
        if self._pk is not None:
            return self._pk
        f = lambda t: t.detach().float().to(_DEV).contiguous()
        h, a = self.hcfg, self.acfg
        F = a["window_size"] // 2 + 1
        ldm = (F + 3) // 4 * 4
        st = self.spectrogram_extractor.stft
        melw = torch.zeros(a["mel_bins"], ldm)
        melw[:, :F] = self.logmel_extractor.melW.detach().float().cpu().t()
        bn = self.bn0
        inv = bn.weight.detach().double().cpu() / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
        db = 10.0 / math.log(10.0)             # 10*log10(x) = db * ln(x): folded into the BatchNorm affine
        pk = dict(dft=ops.pack_conv(torch.cat([st.conv_real.weight[:, 0], st.conv_imag.weight[:, 0]], 0)),
                  mel=ops.pack_conv(melw), F=F, ldm=ldm,
                  bn_scale=f(inv * db), bn_shift=f(bn.bias.detach().double().cpu() - bn.running_mean.detach().double().cpu() * inv),
                  patch=ops.pack_conv(self.patch_embed.proj.weight.reshape(h["embed_dim"], -1), self.patch_embed.proj.bias),
                  patch_ln=(f(self.patch_embed.norm.weight), f(self.patch_embed.norm.bias)),
                  norm=(f(self.norm.weight), f(self.norm.bias)), stages=[])
        dev = pk["bn_scale"].device
        for layer in self.layers:
            blocks = []
            for b in layer.blocks:
                H, W = b.res
                N = b.ws * b.ws
                idx = _window_index(H, W, b.ws, b.shift)
                inv_idx = torch.empty_like(idx)
                inv_idx[idx] = torch.arange(idx.numel())
                at = b.attn
                rb = at.relative_position_bias_table.detach().float().cpu()[at.relative_position_index.view(-1).cpu()]
                rb = rb.view(N, N, -1).permute(2, 0, 1).contiguous()                        # [heads, N, N]
                if b.shift > 0:
                    rb = (rb.unsqueeze(0) + b.attn_mask.detach().float().cpu().unsqueeze(1)).reshape(-1, N, N)  # [nW*heads, N, N]
                blocks.append(dict(
                    heads=b.heads, N=N, shifted=b.shift > 0, nW=(H // b.ws) * (W // b.ws),
                    idx=idx.to(dev), inv=inv_idx.to(dev), bias=f(rb),
                    ln1=(f(b.norm1.weight), f(b.norm1.bias)), ln2=(f(b.norm2.weight), f(b.norm2.bias)),
                    qkv=ops.pack_conv(at.qkv.weight, at.qkv.bias), proj=ops.pack_conv(at.proj.weight, at.proj.bias),
                    fc1=ops.pack_conv(b.mlp.fc1.weight, b.mlp.fc1.bias), fc2=ops.pack_conv(b.mlp.fc2.weight, b.mlp.fc2.bias)))
            st_pk = dict(blocks=blocks, res=layer.blocks[0].res, merge=None)
            if layer.downsample is not None:
                H, W = layer.blocks[0].res
                r2, c2 = torch.meshgrid(torch.arange(H // 2), torch.arange(W // 2), indexing="ij")
                src = torch.stack([(2 * r2) * W + 2 * c2, (2 * r2 + 1) * W + 2 * c2, (2 * r2) * W + 2 * c2 + 1,
                                   (2 * r2 + 1) * W + 2 * c2 + 1], -1)                    # htsat.py:664-668: x0, x1, x2, x3
                st_pk["merge"] = dict(idx=src.reshape(-1).to(dev), ln=(f(layer.downsample.norm.weight), f(layer.downsample.norm.bias)),
                                      red=ops.pack_conv(layer.downsample.reduction.weight, None))
            pk["stages"].append(st_pk)
        self._pk = pk
        return pk

    def _block(self, x, B, L, C, bk):
        """SwinTransformerBlock.forward (htsat.py:556-596) on the window-ordered copy of the tokens."""
        heads, N = bk["heads"], bk["N"]
        d = C // heads
        xw = x.view(B, L, C).index_select(1, bk["idx"])                     # cyclic shift + window partition: one gather
        n = ops.layernorm(xw, *bk["ln1"])                                   # per-token: commutes with the gather
        B_ = B * L // N
        Z = B_ * heads
        qkv = ops.linear(n.view(B_ * N, C), bk["qkv"]).view(B_, N, 3, heads, d)
        q = qkv[:, :, 0].permute(0, 2, 1, 3).reshape(Z, N, d)
        k = qkv[:, :, 1].permute(0, 2, 1, 3).reshape(Z, N, d)
        v = qkv[:, :, 2].permute(0, 2, 1, 3).reshape(Z, N, d)
        s = ops.gemm_nt(q, k, alpha=d ** -0.5)                               # [Z, N, N]
        if bk["shifted"]:                                                   # bias differs per window: fold windows into "heads"
            sv = s.view(B, bk["nW"] * heads, N, N)
        else:
            sv = s.view(B_, heads, N, N)
        ones = torch.ones((sv.shape[0], N), device=x.device, dtype=torch.float32)
        p = ops.softmax_rows_bias(sv, bk["bias"], ones)
        o = ops.gemm_packed_batched(p.view(Z, N, N), ops.pack_kn(v), N, d)
        o = o.view(B_, heads, N, d).permute(0, 2, 1, 3).reshape(B_ * N, C)
        yw = ops.linear(o, bk["proj"], res=xw.view(B_ * N, C))              # shortcut added in window order
        x = yw.view(B, L, C).index_select(1, bk["inv"])                     # window reverse + shift back
        hdn = ops.linear(ops.layernorm(x, *bk["ln2"]).view(B * L, C), bk["fc1"], act=ops.ACT_GELU)
        return ops.linear(hdn, bk["fc2"], res=x.view(B * L, C)).view(B, L, C)

    @torch.no_grad()
    def forward(self, x, mixup_lambda=None, infer_mode=False, device=None):
        """htsat.py:1092-1127 in eval mode, enable_fusion False: x["waveform"] [B, T] at 48 kHz -> {"embedding": [B, 8*embed_dim]}
        (the only output CLAP.get_audio_embedding reads; the classifier outputs are not computed)."""
        pk, h, a = self._prepare(), self.hcfg, self.acfg
        wav = x["waveform"].to(pk["bn_scale"].device).float().contiguous()
        B, T = wav.shape
        n_fft, hop = a["window_size"], a["hop_size"]
        assert T > n_fft // 2, "waveform shorter than the STFT's reflect padding"
        sig = ops.reflect_pad_1d(wav, n_fft // 2)
        frames = T // hop + 1
        spec = ops.frames_gemm(sig, frames, hop, pk["dft"])                          # [B, frames, 2F]
        power = ops.power_spec(spec.view(B * frames, -1), pk["F"], pk["ldm"])
        mel = ops.linear(power, pk["mel"], act=ops.ACT_LOGCLAMP, act_slope=1e-10)    # ln(clamp(power @ melW, amin))
        xm = ops.col_affine(mel, pk["bn_scale"], pk["bn_shift"])                     # -> dB, bn0 (eval)
        S, p = h["spec_size"], h["patch"]
        assert frames <= S * (S // a["mel_bins"]), "the input audio is too long (htsat.py:1069-1071)"
        tok = ops.bicubic_patchify(xm.view(B, frames, a["mel_bins"]), S, p)          # [B, (S/p)^2, p*p]
        xs = ops.layernorm(ops.linear(tok, pk["patch"]), *pk["patch_ln"])
        C = h["embed_dim"]
        for st in pk["stages"]:
            H, W = st["res"]
            L = H * W
            for bk in st["blocks"]:
                xs = self._block(xs, B, L, C, bk)
            if st["merge"] is not None:
                mg = st["merge"]
                xm4 = xs.view(B, L, C).index_select(1, mg["idx"]).view(B, L // 4, 4 * C)
                xs = ops.linear(ops.layernorm(xm4, *mg["ln"]), mg["red"])
                C *= 2
        xs = ops.layernorm(xs, *pk["norm"])
        return {"embedding": ops.token_mean(xs.view(B, -1, C))}


def sinc_resample_kernel(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio.functional.resample's "sinc_interp_hann" kernel (torchaudio==0.13.1 `_get_sinc_resample_kernel`, restated: the
    package's resampler is what encoders/modules.py:700-703 calls): [new, 2*width + orig] fp32 built in fp64."""
    g = math.gcd(orig, new)
    orig, new = orig // g, new // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    k = torch.where(t == 0, torch.ones_like(t), t.sin() / torch.where(t == 0, torch.ones_like(t), t)) * window * (base / orig)
    return k.float().contiguous(), width, orig, new


class CLAPTextModel(nn.Module):
    """clap.open_clip.model.CLAP with a RoBERTa text branch and an HTSAT audio branch: the text side (model.py:513-529, 656-663,
    730-747) and, unless `audio_cfg is False`, the audio side (model.py:563-567, 749-775)."""

    def __init__(self, cfg: dict = None, audio_cfg=None):
        super().__init__()
        self.cfg = dict(cfg or ROBERTA_BASE)
        self.text_branch = _Roberta(self.cfg)
        self.text_projection = nn.Sequential(nn.Linear(self.cfg["hidden_size"], JOINT_DIM), nn.ReLU(),
                                             nn.Linear(JOINT_DIM, JOINT_DIM))
        if audio_cfg is not False:
            self.audio_branch = HTSAT(audio_cfg)
            self.audio_projection = nn.Sequential(nn.Linear(self.audio_branch.num_features, JOINT_DIM), nn.ReLU(),
                                                  nn.Linear(JOINT_DIM, JOINT_DIM))
        else:
            self.audio_branch = None
        self._pk = None
        self._apk = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    def invalidate_packed(self):
        self._pk = None
        self._apk = None
        if self.audio_branch is not None:
            self.audio_branch.invalidate_packed()

    def _prepare(self):
        if self._pk is None:
            f = lambda t: t.detach().float().to(_DEV).contiguous()
            tb = self.text_branch
            layers = []
            for L in tb.encoder.layer:
                a = L.attention
                layers.append(dict(
                    qkv=ops.pack_conv(torch.cat([a.self.query.weight, a.self.key.weight, a.self.value.weight], 0),
                                      torch.cat([a.self.query.bias, a.self.key.bias, a.self.value.bias], 0)),
                    ao=ops.pack_conv(a.output.dense.weight, a.output.dense.bias),
                    ln1=(f(a.output.LayerNorm.weight), f(a.output.LayerNorm.bias)),
                    fi=ops.pack_conv(L.intermediate.dense.weight, L.intermediate.dense.bias),
                    fo=ops.pack_conv(L.output.dense.weight, L.output.dense.bias),
                    ln2=(f(L.output.LayerNorm.weight), f(L.output.LayerNorm.bias))))
            e = tb.embeddings
            self._pk = dict(layers=layers, word=f(e.word_embeddings.weight), pos=f(e.position_embeddings.weight),
                            tok0=f(e.token_type_embeddings.weight[0]), eln=(f(e.LayerNorm.weight), f(e.LayerNorm.bias)),
                            pool=ops.pack_conv(tb.pooler.dense.weight, tb.pooler.dense.bias),
                            p0=ops.pack_conv(self.text_projection[0].weight, self.text_projection[0].bias),
                            p2=ops.pack_conv(self.text_projection[2].weight, self.text_projection[2].bias))
        return self._pk

    @torch.no_grad()
    def get_text_embedding(self, data) -> torch.Tensor:
        """model.py:730-747: {"input_ids", "attention_mask"} [B, T] -> L2-normalised [B, 512]."""
        pk, cfg = self._prepare(), self.cfg
        dev = pk["word"].device
        ids = data["input_ids"].to(dev)
        am = data["attention_mask"].to(dev)
        B, T = ids.shape
        assert T % 4 == 0, "pad the token batch to a multiple of 4 (the reference pads to max_length = 512)"
        H = cfg["num_attention_heads"]
        C = cfg["hidden_size"]
        d = C // H
        m = (ids != cfg["pad_token_id"]).long()
        pos = torch.cumsum(m, dim=1) * m + cfg["pad_token_id"]      # create_position_ids_from_input_ids (host-size index math)
        x = pk["word"].index_select(0, ids.reshape(-1))
        x = ops.axpby(x, pk["pos"].index_select(0, pos.reshape(-1)), 1.0, 1.0)
        x = ops.axpby(x, pk["tok0"].expand(B * T, C).contiguous(), 1.0, 1.0)
        x = ops.layernorm(x, *pk["eln"], eps=cfg["layer_norm_eps"])
        keymask = am.float().contiguous()
        Z = B * H
        for L in pk["layers"]:
            qkv = ops.linear(x, L["qkv"]).view(B, T, 3, H, d)
            q = qkv[:, :, 0].permute(0, 2, 1, 3).reshape(Z, T, d)
            k = qkv[:, :, 1].permute(0, 2, 1, 3).reshape(Z, T, d)
            v = qkv[:, :, 2].permute(0, 2, 1, 3).reshape(Z, T, d)
            s = ops.gemm_nt(q, k, alpha=1.0 / math.sqrt(d))
            p = ops.softmax_rows_masked(s.view(B, H, T, T), keymask, T)      # q_pos0 = T: no causal limit
            o = ops.gemm_packed_batched(p.view(Z, T, T), ops.pack_kn(v), T, d)
            o = o.view(B, H, T, d).permute(0, 2, 1, 3).reshape(B * T, C)
            x = ops.layernorm(ops.linear(o, L["ao"], res=x), *L["ln1"], eps=cfg["layer_norm_eps"])
            h = ops.linear(x, L["fi"], act=ops.ACT_GELU)
            x = ops.layernorm(ops.linear(h, L["fo"], res=x), *L["ln2"], eps=cfg["layer_norm_eps"])
        cls = x.view(B, T, C)[:, 0].contiguous()
        pooled = ops.linear(cls, pk["pool"], act=ops.ACT_TANH)               # RobertaPooler
        e = ops.linear(ops.linear(pooled, pk["p0"], act=ops.ACT_LRELU, act_slope=0.0), pk["p2"])   # :525-529 (ReLU)
        e = e.contiguous()
        return ops.rowscale_add(e, ops.row_l2norm(e, JOINT_DIM), divide=True)  # F.normalize(dim=-1)

    @torch.no_grad()
    def get_audio_embedding(self, data) -> torch.Tensor:
        """model.py:749-775: {"waveform": [B, T] at 48 kHz} -> L2-normalised [B, 512]."""
        if self.audio_branch is None:
            raise RuntimeError("this CLAP was built without its audio branch (audio_cfg=False)")
        if self._apk is None:
            self._apk = (ops.pack_conv(self.audio_projection[0].weight, self.audio_projection[0].bias),
                         ops.pack_conv(self.audio_projection[2].weight, self.audio_projection[2].bias))
        e = self.audio_branch(data)["embedding"]
        e = ops.linear(ops.linear(e, self._apk[0], act=ops.ACT_LRELU, act_slope=0.0), self._apk[1]).contiguous()
        return ops.rowscale_add(e, ops.row_l2norm(e, JOINT_DIM), divide=True)


class CLAPAudioEmbeddingClassifierFreev2(nn.Module):
    def __init__(self, pretrained_path="", enable_cuda=False, sampling_rate=16000, embed_mode="audio", amodel="HTSAT-base",
                 unconditional_prob=0.1, random_mute=False, max_random_mute_portion=0.5, training_mode=True, config=None,
                 audio_config=None):
        """`config` / `audio_config`: geometry overrides for tests (RoBERTa dict / HTSAT dict; audio_config=False builds no
        audio branch); the defaults are roberta-base and HTSAT-base like the reference's create_model(amodel, "roberta")."""
        super().__init__()
        self.device = "cuda"
        self.cuda = enable_cuda
        self.amodel, self.tmodel = amodel, "roberta"
        self.pretrained = pretrained_path
        self.embed_mode = self.embed_mode_orig = embed_mode
        self.sampling_rate = sampling_rate
        self.unconditional_prob = unconditional_prob
        self.random_mute, self.max_random_mute_portion, self.training_mode = random_mute, max_random_mute_portion, training_mode
        self.tokenize = None
        try:  # encoders/modules.py:573; offline the Hub is unreachable
            from transformers import RobertaTokenizer
            self.tokenize = RobertaTokenizer.from_pretrained("roberta-base")
        except Exception:
            pass
        assert amodel == "HTSAT-base", "AudioLDM2 only uses CLAP with the HTSAT-base audio model"
        self.model = CLAPTextModel(config, audio_cfg=audio_config)
        self._resampler = None
        for p in self.model.parameters():
            p.requires_grad = False
        self.unconditional_token = None
        # (global_rows, local_rows) while a prompt-sharded job re-ranks: the reference draws one uniform per row of the GLOBAL
        # candidate batch (audio pass, then text pass); a shard must consume the same draws and keep its rows' decisions
        self.decision_shard = None
        self.weights_loaded = False   # set by _load_from_state_dict: a randomly initialised re-ranker must not rank silently
        self.eval()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """`weights_loaded` = this load supplied every tensor of THIS module (keys under its own prefix).  A post-hook cannot
        tell: PyTorch runs post-hooks on every submodule of a parent `load_state_dict(strict=False)` whether or not a key
        matched, so a checkpoint without `clap.*` would have marked the re-ranker as loaded."""
        own = self.state_dict()
        self.weights_loaded = len(own) > 0 and all((prefix + k) in state_dict for k in own)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def tokenizer(self, text):
        """encoders/modules.py:737-745"""
        if self.tokenize is None:
            raise RuntimeError("CLAP: RobertaTokenizer could not be loaded (no Hub access); tokenize elsewhere and call "
                               "encode_tokens(input_ids, attention_mask)")
        result = self.tokenize(text, padding="max_length", truncation=True, max_length=512, return_tensors="pt")
        return {k: v.squeeze(0) for k, v in result.items()}

    def build_unconditional_emb(self, tokens=None):
        """encoders/modules.py:655-658: the embedding of "" (first of a pair)."""
        data = tokens if tokens is not None else self.tokenizer(["", ""])
        self.unconditional_token = self.model.get_text_embedding(data)[0:1]

    def get_unconditional_condition(self, batchsize):
        """encoders/modules.py:606-610"""
        self.build_unconditional_emb()
        return torch.cat([self.unconditional_token.unsqueeze(0)] * batchsize, dim=0)

    def make_decision(self, probability):
        """encoders/modules.py:618-622: one host-generator uniform per call (RNG contract R2)."""
        return float(torch.rand(1)) < probability

    def encode_tokens(self, input_ids, attention_mask):
        """forward() of "text" mode after the tokenizer (encoders/modules.py:717-735): [B, 1, 512]."""
        if self.unconditional_token is None:
            raise RuntimeError("call build_unconditional_emb(tokens_of_empty_prompt) first (the reference does it lazily "
                               "through its tokenizer, encoders/modules.py:680-681)")
        return self._draw_unconditional(self.model.get_text_embedding({"input_ids": input_ids,
                                                                       "attention_mask": attention_mask}))

    def _draw_unconditional(self, embed):
        """encoders/modules.py:728-735 (both modes; note ddpm.py:114-120 builds the re-ranker with the default probability 0.1)"""
        embed = embed.unsqueeze(1)
        if self.decision_shard is None:
            for i in range(embed.size(0)):
                if self.make_decision(self.unconditional_prob):
                    embed[i] = self.unconditional_token
        else:   # sharded: the global batch's draws, this rank's rows
            g_rows, rows = self.decision_shard
            assert len(rows) == embed.size(0), (len(rows), embed.size(0))
            decisions = [self.make_decision(self.unconditional_prob) for _ in range(g_rows)]
            for i, r in enumerate(rows):
                if decisions[r]:
                    embed[i] = self.unconditional_token
        return embed.detach()

    def encode_audio(self, batch):
        """forward() of "audio" mode (encoders/modules.py:689-716): batch [bs, 1, t] or [bs, t] at self.sampling_rate ->
        [bs, 1, 512].  Resampling follows torchaudio.functional.resample's defaults; `get_audio_features` keeps
        waveform[..., :480000] (clap/training/data.py:421-450; its mel input is unused without fusion)."""
        if self.unconditional_token is None:
            raise RuntimeError("call build_unconditional_emb(tokens_of_empty_prompt) first (the reference does it lazily "
                               "through its tokenizer, encoders/modules.py:680-681)")
        w = torch.as_tensor(batch).to(_DEV).float()
        if w.dim() == 3:
            w = w.squeeze(1)
        w = w.contiguous()
        if self.sampling_rate != 48000:
            if self._resampler is None:
                k, width, down, up = sinc_resample_kernel(self.sampling_rate, 48000)
                self._resampler = (k.to(_DEV), width, down, up)
            k, width, down, up = self._resampler
            w = ops.resample_sinc(w, k, down, up, width, int(math.ceil(up * w.shape[1] / down)))
        w = w[:, :AUDIO_CFG["clip_samples"]].contiguous()
        return self._draw_unconditional(self.model.get_audio_embedding({"waveform": w}))

    def forward(self, batch):
        if self.embed_mode == "audio":
            if self.unconditional_token is None:
                self.build_unconditional_emb()
            return self.encode_audio(batch)
        if self.unconditional_token is None:
            self.build_unconditional_emb()
        text_data = self.tokenizer(batch)
        if isinstance(batch, str) or (isinstance(batch, list) and len(batch) == 1):
            for key in text_data.keys():
                text_data[key] = text_data[key].unsqueeze(0)
        return self.encode_tokens(text_data["input_ids"], text_data["attention_mask"])

    def cos_similarity(self, waveform, text):
        """encoders/modules.py:639-653: waveform [bs, t_steps], text: list of bs prompts — or, where the tokenizer is out of
        reach, the token batch {"input_ids", "attention_mask"} — -> [bs] cosine similarities.  Same order of the host
        generator's draws as the reference: bs for the audio embeddings, then bs for the text embeddings."""
        original_embed_mode = self.embed_mode
        with torch.no_grad():
            self.embed_mode = "audio"
            if isinstance(text, dict):
                audio_emb = self.encode_audio(waveform)
                self.embed_mode = "text"
                text_emb = self.encode_tokens(text["input_ids"], text["attention_mask"])
            else:
                audio_emb = self(waveform)
                self.embed_mode = "text"
                text_emb = self(text)
            similarity = ops.row_cosine(audio_emb[:, 0].contiguous(), text_emb[:, 0].contiguous(), eps=1e-8)
        self.embed_mode = original_embed_mode
        return similarity.squeeze()
