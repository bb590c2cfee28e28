"""KL-VAE (first stage) on MI355X: drop-in for
`audioldm2.latent_encoder.autoencoder.AutoencoderKL` (autoencoder.py:19-117) with the
Encoder/Decoder of `latent_diffusion/modules/diffusionmodules/model.py` (:419-543, :548-686).

Same constructor kwargs (`ddconfig`, `embed_dim`, ...), same state-dict keys
(`encoder.*`, `decoder.*`, `quant_conv.*`, `post_quant_conv.*`, `vocoder.*`), same methods
(`encode(x) -> posterior` with `.sample()`, `decode(z) -> mel`, `.vocoder(mel)`), NCHW fp32 I/O —
usable through `first_stage_config.target` (utils.py:275-310, ddpm.py:766-771).

Execution: channels-last on the HIP implicit-GEMM engine; GroupNorm(eps 1e-6)+swish fused into the
consuming conv, nearest-upsample fused into the up-conv gather; the single-head 4096-token mid
attention is QK^T (NT batched GEMM) -> row softmax -> PV (packed batched GEMM) per sample.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .hifigan import get_vocoder
from .ops import ACT_SILU


def _f(t):
    return t.detach().float().cuda().contiguous()


class ResnetBlock(nn.Module):
    """model.py:122-175 (temb_channels=0 in the VAE)."""

    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=0):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        assert not conv_shortcut and temb_channels == 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = nn.GroupNorm(32, out_channels, eps=1e-6, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)
        self._pk = None

    def run(self, x):
        if self._pk is None:
            self._pk = dict(n1=(_f(self.norm1.weight), _f(self.norm1.bias)),
                            n2=(_f(self.norm2.weight), _f(self.norm2.bias)),
                            c1=ops.pack_conv(self.conv1.weight, self.conv1.bias),
                            c2=ops.pack_conv(self.conv2.weight, self.conv2.bias),
                            nin=ops.pack_conv(self.nin_shortcut.weight, self.nin_shortcut.bias)
                            if self.in_channels != self.out_channels else None)
        pk = self._pk
        if ops.use_dma() and self.in_channels % 32 == 0 and self.out_channels % 32 == 0:
            # round 3: like the UNet's ResBlocks — GroupNorm + swish + operand split once per element (aldm_groupnorm_split),
            # the convs on the DMA-fed kernel over pre-split operands (csrc/igemm_dma.h; bf16x3 products in the default mode)
            if pk["nin"] is None:
                a1, skip = ops.gn_split(x, *pk["n1"], groups=32, eps=1e-6, act=ACT_SILU), x
            else:
                a1, raw = ops.gn_split(x, *pk["n1"], groups=32, eps=1e-6, act=ACT_SILU, want_raw=True)
                skip = ops.conv(raw, pk["nin"])
            h = ops.conv(a1, pk["c1"], pad=(1, 1))
            a2 = ops.gn_split(h, *pk["n2"], groups=32, eps=1e-6, act=ACT_SILU)
            return ops.conv(a2, pk["c2"], pad=(1, 1), res=skip)
        sc, sh = ops.gn_stats(x, *pk["n1"], groups=32, eps=1e-6)
        h = ops.conv(x, pk["c1"], pad=(1, 1), pre=(sc, sh), pre_act=ACT_SILU)
        sc, sh = ops.gn_stats(h, *pk["n2"], groups=32, eps=1e-6)
        skip = x if pk["nin"] is None else ops.conv(x, pk["nin"])
        return ops.conv(h, pk["c2"], pad=(1, 1), pre=(sc, sh), pre_act=ACT_SILU, res=skip)


class AttnBlock(nn.Module):
    """model.py:184-230: single-head attention over H*W tokens, scale C^-0.5."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)
        self._pk = None

    def run(self, x):
        if self._pk is None:
            w = torch.cat([self.q.weight, self.k.weight, self.v.weight], 0)
            b = torch.cat([self.q.bias, self.k.bias, self.v.bias], 0)
            self._pk = dict(n=(_f(self.norm.weight), _f(self.norm.bias)), qkv=ops.pack_conv(w, b),
                            out=ops.pack_conv(self.proj_out.weight, self.proj_out.bias))
        pk = self._pk
        B, H, W, C = x.shape
        L = H * W
        if ops.use_dma() and C % 32 == 0:
            qkv = ops.conv(ops.gn_split(x, *pk["n"], groups=32, eps=1e-6), pk["qkv"]).view(B, L, 3 * C)
        else:
            sc, sh = ops.gn_stats(x, *pk["n"], groups=32, eps=1e-6)
            qkv = ops.conv(x, pk["qkv"], pre=(sc, sh)).view(B, L, 3 * C)
        scores = ops.gemm_nt(qkv[:, :, :C], qkv[:, :, C:2 * C], alpha=float(int(C) ** (-0.5)))
        probs = ops.softmax_rows(scores, 1.0)
        o = ops.gemm_packed_batched(probs, ops.pack_kn(qkv[:, :, 2 * C:]), L, C)
        return ops.conv(o.view(B, H, W, C), pk["out"], res=x)


class _Up(nn.Module):
    """model.py:44-57"""

    def __init__(self, in_channels, with_conv=True):
        super().__init__()
        assert with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)
        self._pk = None

    def run(self, x):
        if self._pk is None:
            self._pk = ops.pack_conv(self.conv.weight, self.conv.bias)
        if ops.use_dma() and x.shape[-1] % 32 == 0:
            x = ops.split_rows(x)   # nearest x2 stays address generation, now over the split image
        return ops.conv(x, self._pk, pad=(1, 1), up=(2, 2))


class _Down(nn.Module):
    """model.py:78-96: F.pad (0,1,0,1) + conv k3 s2 p0 == a gather with zero fill past the edge."""

    def __init__(self, in_channels, with_conv=True):
        super().__init__()
        assert with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 2, 0)
        self._pk = None

    def run(self, x):
        if self._pk is None:
            self._pk = ops.pack_conv(self.conv.weight, self.conv.bias)
        B, H, W, C = x.shape
        xin = ops.split_rows(x) if (ops.use_dma() and C % 32 == 0) else x
        return ops.conv(xin, self._pk, stride=(2, 2), pad=(0, 0), out_hw=((H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1))


def _check_dd(attn_resolutions, downsample_time_stride4_levels, use_linear_attn, attn_type):
    if attn_resolutions or downsample_time_stride4_levels or use_linear_attn or attn_type != "vanilla":
        raise NotImplementedError("VAE(HIP): only the ddconfig option set of the AudioLDM2 configs "
                                  "(utils.py:290-303, 474-487) is implemented")


class Decoder(nn.Module):
    """model.py:548-686"""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions=(), dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False,
                 tanh_out=False, use_linear_attn=False, downsample_time_stride4_levels=(),
                 attn_type="vanilla", **ignorekwargs):
        super().__init__()
        _check_dd(list(attn_resolutions), list(downsample_time_stride4_levels), use_linear_attn, attn_type)
        assert not give_pre_end and not tanh_out
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        block_in = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, dropout=dropout))
                block_in = block_out
            up = nn.Module()
            up.block = block
            up.attn = nn.ModuleList()
            if i_level != 0:
                up.upsample = _Up(block_in, resamp_with_conv)
            self.up.insert(0, up)
        self.norm_out = nn.GroupNorm(32, block_in, eps=1e-6, affine=True)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)
        self._pk = None

    def run(self, z):
        """z: channels-last [B, h, w, z_channels] -> [B, H, W, out_ch]."""
        if self._pk is None:
            self._pk = dict(cin=ops.pack_conv(self.conv_in.weight, self.conv_in.bias),
                            n=(_f(self.norm_out.weight), _f(self.norm_out.bias)),
                            cout=ops.pack_conv(self.conv_out.weight, self.conv_out.bias))
        pk = self._pk
        h = ops.conv(z, pk["cin"], pad=(1, 1))
        h = self.mid.block_1.run(h)
        h = self.mid.attn_1.run(h)
        h = self.mid.block_2.run(h)
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                h = self.up[i_level].block[i_block].run(h)
            if i_level != 0:
                h = self.up[i_level].upsample.run(h)
        sc, sh = ops.gn_stats(h, *pk["n"], groups=32, eps=1e-6)
        return ops.conv(h, pk["cout"], pad=(1, 1), pre=(sc, sh), pre_act=ACT_SILU)


class Encoder(nn.Module):
    """model.py:419-543"""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions=(), dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True,
                 use_linear_attn=False, attn_type="vanilla", downsample_time_stride4_levels=(),
                 **ignore_kwargs):
        super().__init__()
        _check_dd(list(attn_resolutions), list(downsample_time_stride4_levels), use_linear_attn, attn_type)
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block = nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, dropout=dropout))
                block_in = block_out
            down = nn.Module()
            down.block = block
            down.attn = nn.ModuleList()
            if i_level != self.num_resolutions - 1:
                down.downsample = _Down(block_in, resamp_with_conv)
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, dropout=dropout)
        self.norm_out = nn.GroupNorm(32, block_in, eps=1e-6, affine=True)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, 1, 1)
        self.in_channels = in_channels
        self._pk = None

    def run(self, x):
        """x: channels-last [B, T, F, in_channels(=1)].  The 1-channel input conv runs with the single
        channel zero-padded to 4 so the 16-byte gather applies (weights padded to match)."""
        if self._pk is None:
            w = self.conv_in.weight
            if w.shape[1] % 4:
                wp = torch.zeros(w.shape[0], (w.shape[1] + 3) // 4 * 4, 3, 3, dtype=w.dtype, device=w.device)
                wp[:, : w.shape[1]] = w.detach()
                w = wp
            self._pk = dict(cin=ops.pack_conv(w, self.conv_in.bias),
                            n=(_f(self.norm_out.weight), _f(self.norm_out.bias)),
                            cout=ops.pack_conv(self.conv_out.weight, self.conv_out.bias))
        pk = self._pk
        cin_p = pk["cin"].Cin
        if x.shape[-1] != cin_p:
            xp = torch.zeros((*x.shape[:-1], cin_p), device=x.device, dtype=torch.float32)
            xp[..., : x.shape[-1]] = x
            x = xp
        h = ops.conv(x, pk["cin"], pad=(1, 1))
        for i_level in range(self.num_resolutions):
            for i_block in range(self.num_res_blocks):
                h = self.down[i_level].block[i_block].run(h)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample.run(h)
        h = self.mid.block_1.run(h)
        h = self.mid.attn_1.run(h)
        h = self.mid.block_2.run(h)
        sc, sh = ops.gn_stats(h, *pk["n"], groups=32, eps=1e-6)
        return ops.conv(h, pk["cout"], pad=(1, 1), pre=(sc, sh), pre_act=ACT_SILU)


class DiagonalGaussianDistribution(object):
    """modules/distributions/distributions.py:24-41.  `sample()` draws from the CPU generator with
    torch.randn(shape) and moves it to the device — the reference's exact RNG behaviour (contract R)."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape).to(device=self.parameters.device)

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig=None, lossconfig=None, batchsize=None, embed_dim=None, time_shuffle=1,
                 subband=1, sampling_rate=16000, ckpt_path=None, reload_from_ckpt=None, ignore_keys=[],
                 image_key="fbank", colorize_nlabels=None, monitor=None, base_learning_rate=1e-5):
        super().__init__()
        assert "mel_bins" in ddconfig.keys(), "mel_bins is not specified in the Autoencoder config"
        assert ddconfig["double_z"]
        num_mel = ddconfig["mel_bins"]
        self.image_key = image_key
        self.sampling_rate = sampling_rate
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.subband = int(subband)
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        if self.image_key == "fbank":
            self.vocoder = get_vocoder(None, "cpu", num_mel)
        self.embed_dim = embed_dim
        self.time_shuffle = time_shuffle
        self._pk = None
        self.register_load_state_dict_post_hook(lambda module, inc: module.invalidate_packed())
        if ckpt_path is not None:
            sd = torch.load(ckpt_path, map_location="cpu")["state_dict"]
            for k in list(sd.keys()):
                if any(k.startswith(ik) for ik in ignore_keys):
                    del sd[k]
            self.load_state_dict(sd, strict=False)

    def invalidate_packed(self):
        for m in self.modules():
            if hasattr(m, "_pk"):
                m._pk = None

    def _prepare(self):
        if self._pk is None:
            self._pk = dict(post=ops.pack_conv(self.post_quant_conv.weight, self.post_quant_conv.bias),
                            quant=ops.pack_conv(self.quant_conv.weight, self.quant_conv.bias))
        return self._pk

    # -- channels-last fast path (used by the pipeline; no layout round trips) -----------------------
    @torch.no_grad()
    def decode_cl(self, z_cl: torch.Tensor) -> torch.Tensor:
        """z_cl [B, h, w, zc] -> mel channels-last [B, H, W, 1] (== [B, T, F] log-mel)."""
        pk = self._prepare()
        return self.decoder.run(ops.conv(z_cl, pk["post"]))

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """autoencoder.py:111-117: z [B, zc, h, w] -> mel [B, 1, T, F]."""
        if not z.is_cuda:
            raise RuntimeError("AutoencoderKL(HIP) runs on the MI355X only; there is no CPU path")
        dec = self.decode_cl(ops.nchw_to_nhwc(z.float().contiguous()))
        return ops.nhwc_to_nchw(dec)

    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> DiagonalGaussianDistribution:
        """autoencoder.py:103-109: x [B, 1, T, F] -> posterior over [B, zc, T/4, F/4]."""
        if not x.is_cuda:
            raise RuntimeError("AutoencoderKL(HIP) runs on the MI355X only; there is no CPU path")
        pk = self._prepare()
        h = self.encoder.run(ops.nchw_to_nhwc(x.float().contiguous()))
        moments = ops.conv(h, pk["quant"])
        return DiagonalGaussianDistribution(ops.nhwc_to_nchw(moments))
