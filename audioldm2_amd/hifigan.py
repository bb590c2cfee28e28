"""HiFi-GAN generator on MI355X: drop-in for `audioldm2.hifigan.models.Generator` as built by
`get_vocoder` (hifigan/models.py:106-174, utilities/model.py:115-154; weight norm already removed,
so the state dict holds plain `weight` / `bias`).

forward(mel [B, num_mels, T]) -> wave [B, 1, T*prod(upsample_rates)], fp32.
Execution is channels-last ([B, L, C]) on the implicit-GEMM engine:
  * every leaky_relu is fused into the operand gather of the conv that consumes it,
  * ConvTranspose1d runs as `stride` polyphase stride-1 convs writing interleaved rows (no zero
    insertion, no col2im scatter),
  * residual adds, the mean over the parallel ResBlocks (xs/num_kernels) and the final tanh are
    epilogues; dilated convs are the same gather with a tap pitch.
"""
from __future__ import annotations

from typing import Dict

import os

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_LRELU, ACT_TANH

LRELU_SLOPE = 0.1


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


def get_vocoder_config() -> Dict:
    """16 kHz / 64-mel generator hyper-parameters (utilities/model.py:6-37)."""
    return dict(upsample_rates=[5, 4, 2, 2, 2], upsample_kernel_sizes=[16, 16, 8, 4, 4],
                upsample_initial_channel=1024, resblock_kernel_sizes=[3, 7, 11],
                resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], num_mels=64,
                n_fft=1024, hop_size=160, win_size=1024, sampling_rate=16000, fmin=0, fmax=8000)


def get_vocoder_config_48k() -> Dict:
    """48 kHz / 256-mel generator hyper-parameters (utilities/model.py:39-77)."""
    return dict(upsample_rates=[6, 5, 4, 2, 2], upsample_kernel_sizes=[12, 10, 8, 4, 4],
                upsample_initial_channel=1536, resblock_kernel_sizes=[3, 7, 11, 15],
                resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5], [1, 3, 5]], num_mels=256,
                n_fft=2048, hop_size=480, win_size=2048, sampling_rate=48000, fmin=20, fmax=24000)


def _get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


class ResBlock(nn.Module):
    """hifigan/models.py:20-103 parameter holder (convs1: dilated, convs2: dilation 1)."""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.kernel_size = kernel_size
        self.dilation = tuple(dilation)
        self.convs1 = nn.ModuleList([nn.Conv1d(channels, channels, kernel_size, 1, dilation=d,
                                               padding=_get_padding(kernel_size, d)) for d in dilation])
        self.convs2 = nn.ModuleList([nn.Conv1d(channels, channels, kernel_size, 1, dilation=1,
                                               padding=_get_padding(kernel_size, 1)) for _ in dilation])


class Generator(nn.Module):
    def __init__(self, h):
        super().__init__()
        h = AttrDict(h) if not isinstance(h, AttrDict) else h
        self.h = h
        self.num_kernels = len(h.resblock_kernel_sizes)
        self.num_upsamples = len(h.upsample_rates)
        c0 = h.upsample_initial_channel
        self.conv_pre = nn.Conv1d(h.num_mels, c0, 7, 1, padding=3)
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            self.ups.append(nn.ConvTranspose1d(c0 // (2 ** i), c0 // (2 ** (i + 1)), k, u, padding=(k - u) // 2))
        self.resblocks = nn.ModuleList()
        ch = c0
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
                self.resblocks.append(ResBlock(h, ch, k, d))
        self.conv_post = nn.Conv1d(ch, 1, 7, 1, padding=3)
        self._pk = None
        self.register_load_state_dict_post_hook(lambda module, inc: module.invalidate_packed())

    def remove_weight_norm(self):  # API parity with the reference; weights here are already plain
        pass

    def invalidate_packed(self):
        self._pk = None

    def _prepare(self):
        if self._pk is None:
            pk = dict(pre=ops.pack_conv(self.conv_pre.weight, self.conv_pre.bias),
                      post=ops.pack_conv(self.conv_post.weight, self.conv_post.bias), ups=[], res=[])
            for up, u in zip(self.ups, self.h.upsample_rates):
                pk["ups"].append(ops.pack_convtr1d(up.weight, up.bias, u))
            for rb in self.resblocks:
                pk["res"].append(([ops.pack_conv(c.weight, c.bias) for c in rb.convs1],
                                  [ops.pack_conv(c.weight, c.bias) for c in rb.convs2]))
            self._pk = pk
        return self._pk

    @torch.no_grad()
    def forward_cl(self, mel_cl: torch.Tensor) -> torch.Tensor:
        """mel_cl: channels-last [B, T, num_mels] (exactly the VAE decoder's [B, 1, T, F] output) ->
        wave [B, 1, T*hop]."""
        pk = self._prepare()
        B, T, Cm = mel_cl.shape
        x = ops.conv(mel_cl.reshape(B, 1, T, Cm), pk["pre"], pad=(0, 3))
        nk = self.num_kernels
        for i, (u, k) in enumerate(zip(self.h.upsample_rates, self.h.upsample_kernel_sizes)):
            if ops.use_dma() and x.shape[-1] % 32 == 0 and pk["ups"][i][0].N >= self.DMA_MIN_CHANNELS \
                    and pk["ups"][i][0].N % 32 == 0:
                x = self._stage_dma(pk, i, u, k, x)
                continue
            # x = ups[i](leaky_relu(x, 0.1))   models.py:151-152
            p = (k - u) // 2
            L = x.shape[2]
            Lout = (L - 1) * u - 2 * p + k
            phases = pk["ups"][i]
            Tt = phases[0].KW
            y = torch.empty((B, 1, Lout, phases[0].N), device=x.device, dtype=torch.float32)
            Q = (Lout + p) // u + 2
            for ph in range(u):
                ops.conv(x, phases[ph], pad=(0, Tt - 1), out_hw=(1, Q), pre_act=ACT_LRELU,
                         pre_slope=LRELU_SLOPE, out=y, remap=(u, ph - p, Lout))
            x = y
            # xs = sum_j resblock_j(x); x = xs / num_kernels   models.py:153-160
            xs = torch.empty_like(x)
            for j in range(nk):
                rb = self.resblocks[i * nk + j]
                c1s, c2s = pk["res"][i * nk + j]
                kk = rb.kernel_size
                r = x
                nd = len(rb.dilation)
                for m, d in enumerate(rb.dilation):
                    t1 = ops.conv(r, c1s[m], pad=(0, _get_padding(kk, d)), dil=(1, d), pre_act=ACT_LRELU,
                                  pre_slope=LRELU_SLOPE)
                    if m < nd - 1:
                        r = ops.conv(t1, c2s[m], pad=(0, _get_padding(kk, 1)), pre_act=ACT_LRELU,
                                     pre_slope=LRELU_SLOPE, res=r)
                    else:
                        ops.conv(t1, c2s[m], pad=(0, _get_padding(kk, 1)), pre_act=ACT_LRELU,
                                 pre_slope=LRELU_SLOPE, res=r, alpha=1.0 / nk, out=xs, accumulate=(j > 0))
            x = xs
        # x = tanh(conv_post(leaky_relu(x)))  (default slope 0.01, models.py:161-163)
        y = ops.conv(x, pk["post"], pad=(0, 3), pre_act=ACT_LRELU, pre_slope=0.01, act=ACT_TANH)
        return y.view(B, 1, -1)

    # Stages with at least this many output channels run on the DMA-fed GEMM over pre-split operands (round 3): they hold
    # 87 % of the vocoder's FLOPs (L * C^2 per stage: 1.34, 1.34, 0.67, 0.34, 0.17 G for the 16 kHz generator) and are matrix-pipe
    # bound; the 64- / 32-channel stages are HBM bound (K = 3 * C is short) and an extra operand-image pass would cost more
    # than the faster products give back.
    DMA_MIN_CHANNELS = int(os.environ.get("ALDM_HIFIGAN_DMA_MIN", "128"))   # (A/B: tools/hifigan_probe.py)

    def _stage_dma(self, pk, i, u, k, x):
        """One upsampling stage with every leaky_relu applied by the PRODUCER of the conv's operand image instead of the conv's
        gather: split_rows(leaky_relu(x)) in front of the polyphase transposed conv and of the three ResBlocks (one image
        shared by all of them), conv1's epilogue (activation, split image only), conv2's epilogue (fp32 running sum +
        split(leaky_relu(sum)) for the next conv1: out_split_act).  Same arithmetic as the register-staged stage."""
        B = x.shape[0]
        nk = self.num_kernels
        p = (k - u) // 2
        L = x.shape[2]
        Lout = (L - 1) * u - 2 * p + k
        phases = pk["ups"][i]
        Tt = phases[0].KW
        y = torch.empty((B, 1, Lout, phases[0].N), device=x.device, dtype=torch.float32)
        Q = (Lout + p) // u + 2
        xa = ops.split_rows(x, act=ACT_LRELU, slope=LRELU_SLOPE)
        for ph in range(u):
            ops.conv(xa, phases[ph], pad=(0, Tt - 1), out_hw=(1, Q), out=y, remap=(u, ph - p, Lout))
        x = y
        xl = ops.split_rows(x, act=ACT_LRELU, slope=LRELU_SLOPE)
        xs = torch.empty_like(x)
        for j in range(nk):
            rb = self.resblocks[i * nk + j]
            c1s, c2s = pk["res"][i * nk + j]
            kk = rb.kernel_size
            r, rl = x, xl
            nd = len(rb.dilation)
            for m, d in enumerate(rb.dilation):
                t1 = ops.conv(rl, c1s[m], pad=(0, _get_padding(kk, d)), dil=(1, d), act=ACT_LRELU, act_slope=LRELU_SLOPE,
                              split_out="only")
                if m < nd - 1:
                    r, rl = ops.conv(t1, c2s[m], pad=(0, _get_padding(kk, 1)), res=r, split_out="also", split_act=ACT_LRELU,
                                     split_slope=LRELU_SLOPE)
                else:
                    ops.conv(t1, c2s[m], pad=(0, _get_padding(kk, 1)), res=r, alpha=1.0 / nk, out=xs, accumulate=(j > 0))
        return xs

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B, num_mels, T] (reference call convention, ddpm.py:933-935)."""
        if not x.is_cuda:
            raise RuntimeError("Generator(HIP) runs on the MI355X only; there is no CPU path")
        B, Cm, T = x.shape
        cl = ops.nchw_to_nhwc(x.float().contiguous().view(B, Cm, T, 1)).view(B, T, Cm)
        return self.forward_cl(cl)


def get_vocoder(config, device, mel_bins):
    """utilities/model.py:115-154: 64 mel bins -> 16 kHz generator, otherwise the 48 kHz one."""
    cfg = get_vocoder_config() if mel_bins == 64 else get_vocoder_config_48k()
    voc = Generator(AttrDict(cfg))
    voc.eval()
    return voc
