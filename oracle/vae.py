"""Oracle: functional PyTorch-fp32 (CPU) restatement of the KL-VAE decoder (and encoder) and the
HiFi-GAN generator.  TEST INFRA ONLY.

  Decoder.forward      latent_diffusion/modules/diffusionmodules/model.py:653-686
  Encoder.forward      model.py:519-543
  ResnetBlock.forward  model.py:155-175      AttnBlock.forward  model.py:204-230
  Upsample/Downsample  model.py:53-57 / 88-96
  AutoencoderKL.decode latent_encoder/autoencoder.py:111-117 (post_quant_conv then decoder)
  AutoencoderKL.encode autoencoder.py:103-109, DiagonalGaussianDistribution distributions.py:24-41
  Generator.forward    hifigan/models.py:149-165,  ResBlock.forward  hifigan/models.py:96-103
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _gn(x, sd, p):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)  # model.py:38-41


def _conv(x, sd, p, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _swish(x):
    return x * torch.sigmoid(x)  # model.py:33-35


def resnet_block(sd: SD, p: str, x, cin, cout):
    h = _conv(_swish(_gn(x, sd, p + ".norm1")), sd, p + ".conv1", padding=1)
    h = _conv(_swish(_gn(h, sd, p + ".norm2")), sd, p + ".conv2", padding=1)
    if cin != cout:
        x = _conv(x, sd, p + ".nin_shortcut")
    return x + h


def attn_block(sd: SD, p: str, x):
    # model.py:204-230: single head over H*W tokens, scale C^-0.5
    B, C, H, W = x.shape
    h = _gn(x, sd, p + ".norm")
    q = _conv(h, sd, p + ".q").reshape(B, C, H * W).permute(0, 2, 1)
    k = _conv(h, sd, p + ".k").reshape(B, C, H * W)
    v = _conv(h, sd, p + ".v").reshape(B, C, H * W)
    w = torch.bmm(q, k) * (int(C) ** (-0.5))
    w = F.softmax(w, dim=2)
    h = torch.bmm(v, w.permute(0, 2, 1)).reshape(B, C, H, W)
    return x + _conv(h, sd, p + ".proj_out")


@torch.no_grad()
def decoder_forward(sd: SD, dd: dict, z, prefix: str = "decoder."):
    """model.py:653-686 for ddconfig `dd` (attn_resolutions [], no time-stride-4 levels)."""
    ch, mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    nres = len(mult)
    p = prefix
    block_in = ch * mult[-1]
    h = _conv(z, sd, p + "conv_in", padding=1)
    h = resnet_block(sd, p + "mid.block_1", h, block_in, block_in)
    h = attn_block(sd, p + "mid.attn_1", h)
    h = resnet_block(sd, p + "mid.block_2", h, block_in, block_in)
    for lvl in reversed(range(nres)):
        block_out = ch * mult[lvl]
        for ib in range(nrb + 1):
            h = resnet_block(sd, f"{p}up.{lvl}.block.{ib}", h, block_in, block_out)
            block_in = block_out
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(h, sd, f"{p}up.{lvl}.upsample.conv", padding=1)
    h = _swish(_gn(h, sd, p + "norm_out"))
    return _conv(h, sd, p + "conv_out", padding=1)


@torch.no_grad()
def encoder_forward(sd: SD, dd: dict, x, prefix: str = "encoder."):
    """model.py:519-543 (asymmetric (0,1,0,1) pad + stride-2 conv downsample, model.py:88-93)."""
    ch, mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    p = prefix
    h = _conv(x, sd, p + "conv_in", padding=1)
    block_in = ch
    for lvl in range(len(mult)):
        block_out = ch * mult[lvl]
        for ib in range(nrb):
            h = resnet_block(sd, f"{p}down.{lvl}.block.{ib}", h, block_in, block_out)
            block_in = block_out
        if lvl != len(mult) - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = _conv(h, sd, f"{p}down.{lvl}.downsample.conv", stride=2)
    h = resnet_block(sd, p + "mid.block_1", h, block_in, block_in)
    h = attn_block(sd, p + "mid.attn_1", h)
    h = resnet_block(sd, p + "mid.block_2", h, block_in, block_in)
    h = _swish(_gn(h, sd, p + "norm_out"))
    return _conv(h, sd, p + "conv_out", padding=1)


@torch.no_grad()
def vae_decode(sd: SD, dd: dict, z, prefix: str = ""):
    """autoencoder.py:111-117"""
    z = _conv(z, sd, prefix + "post_quant_conv")
    return decoder_forward(sd, dd, z, prefix + "decoder.")


@torch.no_grad()
def vae_encode_moments(sd: SD, dd: dict, x, prefix: str = ""):
    """autoencoder.py:103-109 -> moments [B, 2*z, h, w] (mean | logvar clamped to [-30, 20])"""
    h = encoder_forward(sd, dd, x, prefix + "encoder.")
    return _conv(h, sd, prefix + "quant_conv")


# ---------------------------------------------------------------------------------------------
def hifigan_forward(sd: SD, hcfg: dict, mel, prefix: str = ""):
    """hifigan/models.py:149-165; mel [B, num_mels, T] -> wave [B, 1, T*prod(upsample_rates)].
    Weight norm already removed (utilities/model.py:140), so plain `weight`/`bias` tensors."""
    rates, ksz = hcfg["upsample_rates"], hcfg["upsample_kernel_sizes"]
    rks, rds = hcfg["resblock_kernel_sizes"], hcfg["resblock_dilation_sizes"]
    nk = len(rks)
    p = prefix
    x = F.conv1d(mel, sd[p + "conv_pre.weight"], sd[p + "conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(rates, ksz)):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, sd[f"{p}ups.{i}.weight"], sd[f"{p}ups.{i}.bias"], stride=u,
                               padding=(k - u) // 2)
        xs = None
        for j in range(nk):
            rp = f"{p}resblocks.{i * nk + j}"
            r = x
            for m, d in enumerate(rds[j]):
                kk = rks[j]
                xt = F.leaky_relu(r, 0.1)
                xt = F.conv1d(xt, sd[f"{rp}.convs1.{m}.weight"], sd[f"{rp}.convs1.{m}.bias"],
                              dilation=d, padding=(kk * d - d) // 2)
                xt = F.leaky_relu(xt, 0.1)
                xt = F.conv1d(xt, sd[f"{rp}.convs2.{m}.weight"], sd[f"{rp}.convs2.{m}.bias"],
                              padding=(kk - 1) // 2)
                r = xt + r
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)  # default slope 0.01 (models.py:161)
    x = F.conv1d(x, sd[p + "conv_post.weight"], sd[p + "conv_post.bias"], padding=3)
    return torch.tanh(x)
