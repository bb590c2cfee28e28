"""UNetModel on MI355X: drop-in for the reference's
`audioldm2.latent_diffusion.modules.diffusionmodules.openaimodel.UNetModel` (openaimodel.py:441-885).

Same constructor kwargs, same state-dict keys/shapes (so `load_state_dict(strict=True)` of a reference
checkpoint's `model.diffusion_model.*` entries works), same call signature
`forward(x, timesteps, y=None, context_list=None, context_attn_mask_list=None)` with NCHW fp32 I/O —
usable through the reference's `unet_config.target` plugin seam (utils.py:329-344, ddpm.py:1803).

What is different is everything underneath: the nn.Conv2d/Linear/GroupNorm children are only
parameter holders; forward() runs channels-last on the HIP kernel library (audioldm2_amd.ops):
  * GroupNorm+SiLU are fused into the consuming conv's operand gather (stats kernel + igemm prologue),
  * skip-concat, nearest-upsample, timestep-embedding add and residual add are igemm pro/epilogues,
  * tokens stay [B, H*W, C] so SpatialTransformer needs no rearrange; q/k/v are one fused GEMM,
  * attention is a flash-style fp32 MFMA kernel (no score matrix in HBM).
There is no PyTorch fallback: without libaldm_hip.so forward() raises.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

import os

from . import ops
from .ops import ACT_NONE, ACT_SILU

# A/B switch: ALDM_ATTN_PRESPLIT=0 keeps fp32 K / V and splits them inside the attention kernel (the round-2 path)
# (also off when $ALDM_ATTN_MMA pins the attention kernel to another product mode than the engine's: the images the projection
# writes have the ENGINE's number of parts)
PRESPLIT_ATTENTION = os.environ.get("ALDM_ATTN_PRESPLIT", "1") != "0" and "ALDM_ATTN_MMA" not in os.environ


# ------------------------------------------------------------------------------------------------
# parameter holders (names mirror the reference so checkpoints load unchanged)
# ------------------------------------------------------------------------------------------------
class ResBlock(nn.Module):
    """openaimodel.py:188-300 (use_scale_shift_norm=False, no resblock_updown in AudioLDM2 configs)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(),
                                       nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)
        self._pk = None

    def _prepare(self):
        if self._pk is None:
            c1, c2 = self.in_layers[2], self.out_layers[3]
            self._pk = dict(
                gn1=(self.in_layers[0].weight.detach().float().cuda().contiguous(),
                     self.in_layers[0].bias.detach().float().cuda().contiguous()),
                gn2=(self.out_layers[0].weight.detach().float().cuda().contiguous(),
                     self.out_layers[0].bias.detach().float().cuda().contiguous()),
                conv1=ops.pack_conv(c1.weight, c1.bias),
                conv2=ops.pack_conv(c2.weight, c2.bias),
                skip=None if isinstance(self.skip_connection, nn.Identity)
                else ops.pack_conv(self.skip_connection.weight, self.skip_connection.bias),
            )
        return self._pk

    def run(self, x, e, x2=None):
        """x (++ x2 along C): channels-last [B, H, W, C]; e: [B, out_channels] = this block's slice of
        the batched timestep-embedding projection Linear(SiLU(emb)) (openaimodel.py:244-250, :296-298),
        computed once per forward for all ResBlocks by UNetModel."""
        pk = self._prepare()
        c2 = 0 if x2 is None else x2.shape[-1]
        if ops.use_dma() and self.channels % 32 == 0 and x.shape[-1] % 8 == 0 and c2 % 8 == 0:
            # GroupNorm statistics + apply + SiLU + operand split once per element (aldm_groupnorm_split: one launch up to 1024
            # pixels per sample) instead of once per conv tap in the GEMM's K loop; the convs then run on the DMA-fed kernel
            # over pre-split operands (csrc/igemm_dma.h)
            if pk["skip"] is None:
                assert x2 is None
                a1, skip = ops.gn_split(x, *pk["gn1"], groups=32, eps=1e-5, act=ACT_SILU), x
            else:
                a1, araw = ops.gn_split(x, *pk["gn1"], groups=32, eps=1e-5, x2=x2, act=ACT_SILU, want_raw=True)
                skip = ops.conv(araw, pk["skip"])
            h = ops.conv(a1, pk["conv1"], pad=(1, 1), rowbias=e)
            a2 = ops.gn_split(h, *pk["gn2"], groups=32, eps=1e-5, act=ACT_SILU)
            return ops.conv(a2, pk["conv2"], pad=(1, 1), res=skip)
        sc, sh = ops.gn_stats(x, *pk["gn1"], groups=32, eps=1e-5, x2=x2)
        h = ops.conv(x, pk["conv1"], x2=x2, pad=(1, 1), pre=(sc, sh), pre_act=ACT_SILU, rowbias=e)
        sc2, sh2 = ops.gn_stats(h, *pk["gn2"], groups=32, eps=1e-5)
        if pk["skip"] is None:
            assert x2 is None
            skip = x
        else:
            skip = ops.conv(x, pk["skip"], x2=x2)
        return ops.conv(h, pk["conv2"], pad=(1, 1), pre=(sc2, sh2), pre_act=ACT_SILU, res=skip)


class Downsample(nn.Module):
    """openaimodel.py:150-186 (conv_resample=True): conv3x3 stride 2."""

    def __init__(self, channels, use_conv=True, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert use_conv and dims == 2
        self.channels = channels
        self.out_channels = out_channels or channels
        self.op = nn.Conv2d(channels, self.out_channels, 3, stride=2, padding=padding)
        self._pk = None

    def run(self, x):
        if self._pk is None:
            self._pk = ops.pack_conv(self.op.weight, self.op.bias)
        if ops.use_dma() and self.channels % 32 == 0:
            x = ops.split_rows(x)
        return ops.conv(x, self._pk, stride=(2, 2), pad=(1, 1))


class Upsample(nn.Module):
    """openaimodel.py:106-136: nearest x2 then conv3x3 — fused into one gather."""

    def __init__(self, channels, use_conv=True, dims=2, out_channels=None, padding=1):
        super().__init__()
        assert use_conv and dims == 2
        self.channels = channels
        self.out_channels = out_channels or channels
        self.conv = nn.Conv2d(channels, self.out_channels, 3, padding=padding)
        self._pk = None

    def run(self, x):
        if self._pk is None:
            self._pk = ops.pack_conv(self.conv.weight, self.conv.bias)
        if ops.use_dma() and self.channels % 32 == 0:
            x = ops.split_rows(x)
        return ops.conv(x, self._pk, pad=(1, 1), up=(2, 2))


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    """attention.py:47-63 (glu=True, mult=4)"""

    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * mult), nn.Dropout(dropout), nn.Linear(dim * mult, dim))


class CrossAttention(nn.Module):
    """attention.py:325-367 parameter holder."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        context_dim = query_dim if context_dim is None else context_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))


class BasicTransformerBlock(nn.Module):
    """attention.py:370-410"""

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None):
        super().__init__()
        assert d_head == 32, "the HIP attention kernel is specialised for head dim 32 (all AudioLDM2 configs)"
        self.attn1 = CrossAttention(dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout)
        self.attn2 = CrossAttention(dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.heads = n_heads
        self.dim = dim
        self._pk = None
        self._kv = None  # [(context tensor, its _version, K|V projection)], newest first — see _context_kv

    def _context_kv(self, context, pk):
        """K/V projection of the cross-attention context (attention.py:336-337).  The context is the
        same tensor for all 200 DDIM steps, so the projection is computed once and reused while the
        SAME tensor object (unmodified: same _version) is passed again.  The cache holds a reference to
        the context, so its storage cannot be recycled under us."""
        for c in (self._kv or ()):
            if c[0] is context and c[1] == context._version:
                return c[2]
        kv = ops.linear(context, pk["kv2"])
        if not torch.cuda.is_current_stream_capturing():  # graph-pool memory must not outlive the graph
            # two entries: the uncond and the cond context when CFG runs as two concurrent half passes
            self._kv = [(context, context._version, kv)] + list(self._kv or ())[:1]
        return kv

    def _prepare(self):
        if self._pk is None:
            f = lambda t: t.detach().float().cuda().contiguous()
            a1, a2 = self.attn1, self.attn2
            self._pk = dict(
                ln=[(f(n.weight), f(n.bias)) for n in (self.norm1, self.norm2, self.norm3)],
                qkv1=ops.pack_conv(torch.cat([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight], 0)),
                out1=ops.pack_conv(a1.to_out[0].weight, a1.to_out[0].bias),
                # attn2 used as self-attention when no context is routed to this transformer
                qkv2=ops.pack_conv(torch.cat([a2.to_q.weight, a2.to_k.weight, a2.to_v.weight], 0))
                if a2.to_k.weight.shape[1] == self.dim else None,
                q2=ops.pack_conv(a2.to_q.weight),
                kv2=ops.pack_conv(torch.cat([a2.to_k.weight, a2.to_v.weight], 0)),
                out2=ops.pack_conv(a2.to_out[0].weight, a2.to_out[0].bias),
                ff1=ops.pack_geglu(self.ff.net[0].proj.weight, self.ff.net[0].proj.bias),
                ff2=ops.pack_conv(self.ff.net[2].weight, self.ff.net[2].bias),
            )
        return self._pk

    def run(self, h, context=None, mask=None, want_split=False):
        """h: [B, L, C] tokens.  attention.py:406-410 (mask is ignored when context is None: :400-404).
        want_split: also return the result as a split image (the operand of SpatialTransformer.proj_out)."""
        pk = self._prepare()
        C = self.dim
        # DMA mode: every GEMM operand is written pre-split by its producer (LayerNorm, attention, GEGLU epilogue)
        so = "only" if (ops.use_dma() and C % 32 == 0) else None
        n = ops.layernorm(h, *pk["ln"][0], split_out=so)
        # self-attention over operands the projection's epilogue pre-splits (round 3): k as a split image, v transposed per key
        # tile — when the token count is a whole number of 32-key tiles (every UNet level of every config)
        # (... and the width is a whole number of 64-column tiles — an even head count — and the projection has no bias: what
        # ALDM_EPI_QKV requires; anything else takes the fp32 K / V path)
        pre = so is not None and PRESPLIT_ATTENTION and h.shape[1] % 32 == 0 and C % 64 == 0 and pk["qkv1"].bias is None
        if pre:
            q, kimg, vtimg = ops.linear_qkv(n, pk["qkv1"], self.heads, h.shape[1])
            a = ops.attention_presplit(q, kimg, vtimg, self.heads, split_out=so)
        else:
            qkv = ops.linear(n, pk["qkv1"])
            a = ops.attention(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], self.heads, split_out=so)
        h = ops.linear(a, pk["out1"], res=h)
        n = ops.layernorm(h, *pk["ln"][1], split_out=so)
        if context is None:
            if pk["qkv2"] is None:
                raise RuntimeError("attn2 was built with a context_dim but no context was provided")
            if pre and pk["qkv2"].bias is None:
                q, kimg, vtimg = ops.linear_qkv(n, pk["qkv2"], self.heads, h.shape[1])
                a = ops.attention_presplit(q, kimg, vtimg, self.heads, split_out=so)
            else:
                qkv = ops.linear(n, pk["qkv2"])
                a = ops.attention(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], self.heads, split_out=so)
        else:
            q = ops.linear(n, pk["q2"])
            kv = self._context_kv(context, pk)
            a = ops.attention(q, kv[:, :, :C], kv[:, :, C:], self.heads, mask=mask, split_out=so)
        h = ops.linear(a, pk["out2"], res=h)
        n = ops.layernorm(h, *pk["ln"][2], split_out=so)
        g = ops.linear_geglu(n, pk["ff1"], split_out=so)  # Linear(C, 8C) + x*gelu(gate) in one GEMM (attention.py:37-45)
        if want_split and so:
            return ops.linear(g, pk["ff2"], res=h, split_out="also")
        return ops.linear(g, pk["ff2"], res=h)


class SpatialTransformer(nn.Module):
    """attention.py:413-467"""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None):
        super().__init__()
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim)
             for _ in range(depth)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)
        self._pk = None

    def run(self, x, context=None, mask=None):
        if self._pk is None:
            self._pk = dict(gn=(self.norm.weight.detach().float().cuda().contiguous(),
                                self.norm.bias.detach().float().cuda().contiguous()),
                            pin=ops.pack_conv(self.proj_in.weight, self.proj_in.bias),
                            pout=ops.pack_conv(self.proj_out.weight, self.proj_out.bias))
        pk = self._pk
        B, H, W, C = x.shape
        if ops.use_dma() and C % 32 == 0 and self.proj_in.out_channels % 32 == 0:
            h = ops.conv(ops.gn_split(x, *pk["gn"], groups=32, eps=1e-6), pk["pin"]).view(B, H * W, -1)
            hs = None
            for i, blk in enumerate(self.transformer_blocks):
                r = blk.run(h, context, mask, want_split=i == len(self.transformer_blocks) - 1)
                h, hs = r if isinstance(r, tuple) else (r, None)
            return ops.conv((hs if hs is not None else h).view(B, H, W, -1), pk["pout"], res=x)
        sc, sh = ops.gn_stats(x, *pk["gn"], groups=32, eps=1e-6)
        h = ops.conv(x, pk["pin"], pre=(sc, sh)).view(B, H * W, -1)
        for blk in self.transformer_blocks:
            h = blk.run(h, context, mask)
        return ops.conv(h.view(B, H, W, -1), pk["pout"], res=x)


class TimestepEmbedSequential(nn.Sequential):
    """openaimodel.py:75-103: routes emb to ResBlocks and (context, mask) to SpatialTransformers; the
    first transformer of a block never gets a context, transformers beyond the list get None."""

    def run(self, x, emb, context_list, mask_list, x2=None, lo=0, hi=None):
        """emb: dict-like indexable by a ResBlock's `_emb_slice` -> [B, out_channels] row-bias view.  lo / hi: run layers
        [lo, hi) only (UNetModel's shared classifier-free-guidance prefix stops inside a block)."""
        ctxs = [None] + list(context_list)
        masks = [None] + list(mask_list)
        layers = list(self)
        st_id = sum(isinstance(l, SpatialTransformer) for l in layers[:lo])
        for layer in layers[lo:hi]:
            if isinstance(layer, ResBlock):
                x = layer.run(x, emb[layer._emb_slice], x2)
                x2 = None
            elif isinstance(layer, SpatialTransformer):
                if st_id >= len(ctxs):
                    c, m = None, None
                else:
                    c, m = ctxs[st_id], masks[st_id]
                x = layer.run(x, c, m)
                st_id += 1
            elif isinstance(layer, (Downsample, Upsample)):
                x = layer.run(x)
            elif isinstance(layer, _ConvIn):
                x = layer.run(x)
            else:  # pragma: no cover
                raise RuntimeError(f"unexpected layer {type(layer)}")
        assert x2 is None, "skip tensor was not consumed by a ResBlock"
        return x


class _EmbSlices:
    """Column-slice views of the batched timestep-embedding projection [B, sum(N)]."""

    def __init__(self, t):
        self.t = t

    def __getitem__(self, sl):
        return self.t[:, sl]


class _ConvIn(nn.Conv2d):
    """input_blocks.0.0: plain conv3x3 (openaimodel.py:570-574) — subclass only to carry run()."""

    _pk = None

    def run(self, x):
        if self._pk is None:
            self._pk = ops.pack_conv(self.weight, self.bias)
        return ops.conv(x, self._pk, pad=(1, 1))


# ------------------------------------------------------------------------------------------------
class UNetModel(nn.Module):
    """Constructor kwargs as openaimodel.py:469-497; see module docstring."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2,
                 extra_sa_layer=True, num_classes=None, extra_film_condition_dim=None,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=True, transformer_depth=1,
                 context_dim=None, n_embed=None, legacy=True):
        super().__init__()
        if dims != 2 or use_scale_shift_norm or resblock_updown or not use_spatial_transformer \
                or n_embed is not None or num_classes is not None or use_fp16 or not conv_resample:
            raise NotImplementedError("UNetModel(HIP): only the option set used by the AudioLDM2 configs "
                                      "(utils.py:329-343) is implemented")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        assert num_heads != -1 or num_head_channels != -1, "Either num_heads or num_head_channels has to be set"
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.channel_mult = channel_mult
        self.extra_film_condition_dim = extra_film_condition_dim
        self.use_extra_film_by_concat = extra_film_condition_dim is not None
        self.dtype = torch.float32
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        if self.use_extra_film_by_concat:
            self.film_emb = nn.Linear(extra_film_condition_dim, ted)
        if context_dim is not None and not isinstance(context_dim, (list, tuple)):
            context_dim = [context_dim]
        elif context_dim is None:
            context_dim = [None]
        context_dim = list(context_dim)
        emb_ch = ted * 2 if self.use_extra_film_by_concat else ted

        def transformers(ch):
            if num_head_channels == -1:
                heads, dim_head = num_heads, ch // num_heads
            else:
                heads, dim_head = ch // num_head_channels, num_head_channels
            out = []
            if extra_sa_layer:
                out.append(SpatialTransformer(ch, heads, dim_head, depth=transformer_depth, context_dim=None))
            for cd in context_dim:
                out.append(SpatialTransformer(ch, heads, dim_head, depth=transformer_depth, context_dim=cd))
            return out

        self.input_blocks = nn.ModuleList(
            [TimestepEmbedSequential(_ConvIn(in_channels, model_channels, 3, padding=1))])
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, emb_ch, dropout, out_channels=mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers += transformers(ch)
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, True, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(
            ResBlock(ch, emb_ch, dropout), *transformers(ch), ResBlock(ch, emb_ch, dropout))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [ResBlock(ch + ich, emb_ch, dropout, out_channels=model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers += transformers(ch)
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, True, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(),
                                 nn.Conv2d(model_channels, out_channels, 3, padding=1))
        self._pk = None
        self._graph_cache = {}  # captured DDIM step graph + its static buffers (ddim.DDIMSampler)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    # -- packed-weight cache ---------------------------------------------------------------------
    def invalidate_packed(self):
        """Drop every re-laid-out weight copy, cached context projection and captured step graph (call
        after mutating parameters)."""
        from .ddim import drop_graph_entries
        drop_graph_entries(self._graph_cache)   # frees the graphs now, not at some later cyclic collection
        for m in self.modules():
            if hasattr(m, "_pk"):
                m._pk = None
            if hasattr(m, "_kv"):
                m._kv = None

    def drop_step_caches(self):
        """Drop what depends on the matrix-core mode but not on the weights: the captured step graph and the cached
        cross-attention K/V projections (call after ops.set_mma(); the packed weights keep both split images)."""
        from .ddim import drop_graph_entries
        drop_graph_entries(self._graph_cache)
        for m in self.modules():
            if hasattr(m, "_kv"):
                m._kv = None

    def collect_context_kv(self, contexts):
        """[(block, context, kv)] for every cached cross-attention K/V projection of the given context tensors.  A
        captured step graph reads exactly these kv buffers: the graph-cache entry keeps this list (and with it the
        buffers) alive, so nothing a later run does to the blocks' 2-entry lookup lists can free or recycle them."""
        out = []
        for m in self.modules():
            if isinstance(m, BasicTransformerBlock) and m._kv:
                for ctx, _ver, kv in m._kv:
                    if any(ctx is c for c in contexts):
                        out.append((m, ctx, kv))
        return out

    def refresh_context_kv(self, entries):
        """Recompute IN PLACE the K/V projections a captured step graph reads (`entries` from collect_context_kv; the
        contexts' contents were overwritten for a new sampling run) and put them back at the front of each block's
        lookup list.  Returns the number of projections refreshed."""
        n = 0
        for m, ctx, kv in entries:
            pk = m._prepare()
            ops.linear(ctx, pk["kv2"], out=kv)
            m._kv = [(ctx, ctx._version, kv)] + [c for c in (m._kv or ()) if c[2] is not kv][:1]
            n += 1
        return n

    def _prepare(self):
        if self._pk is None:
            f = lambda t: t.detach().float().cuda().contiguous()
            # every ResBlock's emb_layers Linear, concatenated along N: ONE [B, emb] x [emb, sum(N)] GEMM
            # per forward instead of 22 tiny launches; block i reads columns _emb_slice
            rbs = [m for m in self.modules() if isinstance(m, ResBlock)]
            off = 0
            for m in rbs:
                m._emb_slice = slice(off, off + m.out_channels)
                off += m.out_channels
            self._pk = dict(
                emb_all=ops.pack_conv(torch.cat([m.emb_layers[1].weight for m in rbs], 0),
                                      torch.cat([m.emb_layers[1].bias for m in rbs], 0)),
                te0=ops.pack_conv(self.time_embed[0].weight, self.time_embed[0].bias),
                te2=ops.pack_conv(self.time_embed[2].weight, self.time_embed[2].bias),
                film=ops.pack_conv(self.film_emb.weight, self.film_emb.bias)
                if self.use_extra_film_by_concat else None,
                gn=(f(self.out[0].weight), f(self.out[0].bias)),
                out=ops.pack_conv(self.out[2].weight, self.out[2].bias),
            )
        return self._pk

    # -- forward -----------------------------------------------------------------------------------
    def _shared_prefix_end(self, context_list):
        """(input block, layer) of the first SpatialTransformer that receives a context — where the two halves of a
        classifier-free-guidance batch start to differ (TimestepEmbedSequential: the first transformer of a block never gets a
        context, transformer k > 0 gets context_list[k - 1]); None when the halves differ from the start (FiLM: `y` enters every
        ResBlock) or no transformer ever gets a context."""
        if self.use_extra_film_by_concat:
            return None
        for bi, blk in enumerate(self.input_blocks):
            st = 0
            for li, layer in enumerate(blk):
                if isinstance(layer, SpatialTransformer):
                    if st >= 1 and st - 1 < len(context_list) and context_list[st - 1] is not None:
                        return bi, li
                    st += 1
        return None

    @torch.no_grad()
    def forward(self, x, timesteps=None, y=None, context_list=None, context_attn_mask_list=None, cfg_shared=False, **kwargs):
        """openaimodel.py:837-885.  x: [N, C, H, W] fp32 on the GPU; returns eps [N, C_out, H, W] (cfg_shared: x [B, ...] once for
        the 2B rows [uncond ; cond] of a classifier-free-guidance pass, returns eps [2B, ...]; see below)."""
        assert (y is not None) == self.use_extra_film_by_concat, \
            "must specify y if and only if the model is class-conditional or film embedding conditional"
        if not x.is_cuda:
            raise RuntimeError("UNetModel(HIP) runs on the MI355X only; there is no CPU path")
        pk = self._prepare()
        context_list = [c.float().contiguous() if c is not None else None for c in (context_list or [])]
        mask_list = list(context_attn_mask_list or [])
        # Shared classifier-free-guidance prefix (round 5).  With cfg_shared=True the caller passes x ONCE ([B, ...]) for a batch
        # whose rows [0, B) (unconditional) and [B, 2B) (conditional) see the same x and the same t and differ only in their
        # contexts (ddim.py:293-296 runs them as two passes over the same x).  Until the first SpatialTransformer that RECEIVES a
        # context the two halves compute identical values — conv_in, the level-0 ResBlocks, the first Downsample, the first
        # level-1 ResBlock, the context-free first transformer — so that prefix runs on B samples and its output (and its skip
        # tensors, when the decoder pops them) is duplicated: 4 of 7 level-0 convs at half the rows, ~0.5 ms of a 22 ms step.
        div = self._shared_prefix_end(context_list) if cfg_shared else None
        B = x.shape[0]
        if cfg_shared:
            assert timesteps.shape[0] in (B, 2 * B) and all(c is None or c.shape[0] == 2 * B for c in context_list), \
                "cfg_shared: x [B, ...] once, contexts for the 2B rows [uncond ; cond]"
            timesteps = timesteps[:B]
            if div is None:   # FiLM-conditioned model (y differs between the halves from the first ResBlock on): nothing to share
                x = x.repeat(2, 1, 1, 1)
                timesteps = timesteps.repeat(2)
        t_emb = ops.timestep_embedding(timesteps, self.model_channels)
        emb = ops.linear(ops.linear(t_emb, pk["te0"], act=ACT_SILU), pk["te2"])
        if self.use_extra_film_by_concat:
            emb = torch.cat([emb, ops.linear(y.float().contiguous(), pk["film"])], dim=-1).contiguous()
        emb_rows = ops.linear(emb, pk["emb_all"], pre_act=ACT_SILU)
        emb = _EmbSlices(emb_rows)
        h = ops.nchw_to_nhwc(x.float().contiguous())
        hs = []
        if div is None:
            for module in self.input_blocks:
                h = module.run(h, emb, context_list, mask_list)
                hs.append(h)
        else:
            dup = lambda t: t.repeat(2, *([1] * (t.dim() - 1)))   # rows [B, 2B) = rows [0, B)
            emb_half, emb = emb, _EmbSlices(dup(emb_rows))
            for bi, module in enumerate(self.input_blocks):
                if bi < div[0]:
                    h = module.run(h, emb_half, context_list, mask_list)
                elif bi == div[0]:
                    h = module.run(h, emb_half, context_list, mask_list, hi=div[1])
                    h = module.run(dup(h), emb, context_list, mask_list, lo=div[1])
                else:
                    h = module.run(h, emb, context_list, mask_list)
                hs.append(h)
        h = self.middle_block.run(h, emb, context_list, mask_list)
        for module in self.output_blocks:
            skip = hs.pop()
            if skip.shape[0] != h.shape[0]:   # a skip tensor of the shared prefix
                skip = dup(skip)
            h = module.run(h, emb, context_list, mask_list, x2=skip)
        sc, sh = ops.gn_stats(h, *pk["gn"], groups=32, eps=1e-5)
        out = ops.conv(h, pk["out"], pad=(1, 1), pre=(sc, sh), pre_act=ACT_SILU)
        return ops.nhwc_to_nchw(out)
