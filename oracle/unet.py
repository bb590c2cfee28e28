"""Oracle: functional PyTorch-fp32 (CPU) restatement of the reference UNet denoiser.  TEST INFRA ONLY.

Restates `UNetModel.forward` (latent_diffusion/modules/diffusionmodules/openaimodel.py:837-885) and
everything it reaches, as pure functions over a state dict that uses the reference's parameter names:
  ResBlock._forward            openaimodel.py:280-300
  Downsample / Upsample        openaimodel.py:184-186 / 126-136
  TimestepEmbedSequential      openaimodel.py:81-103  (context routing: 1st transformer gets None)
  SpatialTransformer.forward   modules/attention.py:456-467
  BasicTransformerBlock        modules/attention.py:400-410 (mask is dropped when context is None)
  CrossAttention.forward       modules/attention.py:343-367
  GEGLU / FeedForward          modules/attention.py:37-63
  timestep_embedding           diffusionmodules/util.py:172-196
Pinned against the real classes by tests/golden (oracle/make_golden.py) and, when /root/reference
exists, tests/test_oracle_vs_reference.py.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    # util.py:172-196: [cos | sin], freqs = exp(-ln(max_period) * i / half)
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def unet_layout(cfg: dict):
    """Enumerate the module structure built by UNetModel.__init__ (openaimodel.py:476-832):
    returns (input_blocks, middle, output_blocks), each a list of blocks, each block a list of
    ("conv"|"res"|"st"|"down"|"up", params...) tuples in module order."""
    mc = cfg["model_channels"]
    mult = list(cfg.get("channel_mult", (1, 2, 4, 8)))
    nrb = cfg["num_res_blocks"]
    att = set(cfg["attention_resolutions"])
    ctx = cfg.get("context_dim", None)
    if ctx is not None and not isinstance(ctx, (list, tuple)):
        ctx = [ctx]
    elif ctx is None:
        ctx = [None]
    ctx = list(ctx)
    nhc = cfg.get("num_head_channels", -1)
    nheads = cfg.get("num_heads", -1)
    extra_sa = cfg.get("extra_sa_layer", True)
    depth = cfg.get("transformer_depth", 1)

    def heads_for(ch):
        if nhc == -1:
            return nheads, ch // nheads
        return ch // nhc, nhc  # legacy=True + spatial transformer: dim_head = ch // num_heads

    def sts(ch):
        h, d = heads_for(ch)
        out = []
        if extra_sa:
            out.append(("st", ch, h, d, None, depth))
        for c in ctx:
            out.append(("st", ch, h, d, c, depth))
        return out

    inp = [[("conv", cfg["in_channels"], mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            blk = [("res", ch, m * mc)]
            ch = m * mc
            if ds in att:
                blk += sts(ch)
            inp.append(blk)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    mid = [("res", ch, ch)] + sts(ch) + [("res", ch, ch)]
    outb = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            blk = [("res", ch + ich, mc * m)]
            ch = mc * m
            if ds in att:
                blk += sts(ch)
            if level and i == nrb:
                blk.append(("up", ch))
                ds //= 2
            outb.append(blk)
    return inp, mid, outb


def _gn(x, sd, p, eps):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(x, sd, p, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def resblock(sd: SD, p: str, x, emb, cin, cout):
    # openaimodel.py:280-300 (no up/down, no scale-shift norm in AudioLDM2 configs)
    h = _conv(F.silu(_gn(x, sd, p + ".in_layers.0", 1e-5)), sd, p + ".in_layers.2", padding=1)
    e = _lin(F.silu(emb), sd, p + ".emb_layers.1")
    h = h + e[:, :, None, None]
    h = _conv(F.silu(_gn(h, sd, p + ".out_layers.0", 1e-5)), sd, p + ".out_layers.3", padding=1)
    skip = x if cin == cout else _conv(x, sd, p + ".skip_connection")
    return skip + h


def cross_attention(sd: SD, p: str, x, heads, context=None, mask=None):
    # attention.py:343-367
    B, L, _ = x.shape
    ctx = x if context is None else context
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    d = q.shape[-1] // heads

    def split(t):
        return t.view(B, -1, heads, d).permute(0, 2, 1, 3).reshape(B * heads, -1, d)

    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("bid,bjd->bij", q, k) * (d ** -0.5)
    if mask is not None:
        m = mask.reshape(B, -1)
        m = m[:, None, :].expand(B, heads, m.shape[-1]).reshape(B * heads, 1, -1)
        sim = sim.masked_fill(~(m == 1), -torch.finfo(sim.dtype).max)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", attn, v)
    out = out.view(B, heads, L, d).permute(0, 2, 1, 3).reshape(B, L, heads * d)
    return _lin(out, sd, p + ".to_out.0")


def transformer_block(sd: SD, p: str, x, heads, context=None, mask=None):
    # attention.py:400-410; BasicTransformerBlock.forward drops the mask when context is None
    if context is None:
        mask = None
    x = cross_attention(sd, p + ".attn1", F.layer_norm(x, x.shape[-1:], sd[p + ".norm1.weight"], sd[p + ".norm1.bias"]), heads) + x
    x = cross_attention(sd, p + ".attn2", F.layer_norm(x, x.shape[-1:], sd[p + ".norm2.weight"], sd[p + ".norm2.bias"]), heads,
                        context=context, mask=mask) + x
    h = F.layer_norm(x, x.shape[-1:], sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])
    h = _lin(h, sd, p + ".ff.net.0.proj")
    a, gate = h.chunk(2, dim=-1)
    h = _lin(a * F.gelu(gate), sd, p + ".ff.net.2")
    return h + x


def spatial_transformer(sd: SD, p: str, x, heads, depth, context=None, mask=None):
    # attention.py:456-467
    B, C, H, W = x.shape
    x_in = x
    h = _conv(_gn(x, sd, p + ".norm", 1e-6), sd, p + ".proj_in")
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, -1)
    for d in range(depth):
        h = transformer_block(sd, f"{p}.transformer_blocks.{d}", h, heads, context, mask)
    h = h.reshape(B, H, W, -1).permute(0, 3, 1, 2)
    return _conv(h, sd, p + ".proj_out") + x_in


def _run_block(sd: SD, p: str, blk, h, emb, context_list, mask_list):
    # TimestepEmbedSequential.forward, openaimodel.py:81-103
    ctxs = [None] + list(context_list)
    masks = [None] + list(mask_list)
    st_id = 0
    for j, layer in enumerate(blk):
        kind = layer[0]
        lp = f"{p}.{j}"
        if kind == "conv":
            h = _conv(h, sd, lp, padding=1)
        elif kind == "res":
            h = resblock(sd, lp, h, emb, layer[1], layer[2])
        elif kind == "st":
            if st_id >= len(ctxs):
                c, m = None, None
            else:
                c, m = ctxs[st_id], masks[st_id]
            h = spatial_transformer(sd, lp, h, layer[2], layer[5], c, m)
            st_id += 1
        elif kind == "down":
            h = _conv(h, sd, lp + ".op", stride=2, padding=1)  # openaimodel.py:172,184-186
        elif kind == "up":
            h = _conv(F.interpolate(h, scale_factor=2, mode="nearest"), sd, lp + ".conv", padding=1)
    return h


@torch.no_grad()
def unet_forward(sd: SD, cfg: dict, x, timesteps, context_list: Optional[List] = None,
                 context_attn_mask_list: Optional[List] = None, y=None, prefix: str = ""):
    """openaimodel.py:837-885.  x: [B, C, H, W] fp32; timesteps: [B]; returns eps [B, C_out, H, W]."""
    context_list = list(context_list or [])
    mask_list = list(context_attn_mask_list or [])
    inp, mid, outb = unet_layout(cfg)
    pre = prefix
    t_emb = timestep_embedding(timesteps, cfg["model_channels"])
    emb = _lin(F.silu(_lin(t_emb, sd, pre + "time_embed.0")), sd, pre + "time_embed.2")
    if cfg.get("extra_film_condition_dim") is not None:
        emb = torch.cat([emb, _lin(y, sd, pre + "film_emb")], dim=-1)  # openaimodel.py:869-870
    hs = []
    h = x.float()
    for i, blk in enumerate(inp):
        h = _run_block(sd, f"{pre}input_blocks.{i}", blk, h, emb, context_list, mask_list)
        hs.append(h)
    h = _run_block(sd, f"{pre}middle_block", mid, h, emb, context_list, mask_list)
    for i, blk in enumerate(outb):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_block(sd, f"{pre}output_blocks.{i}", blk, h, emb, context_list, mask_list)
    h = F.silu(_gn(h, sd, pre + "out.0", 1e-5))
    return _conv(h, sd, pre + "out.2", padding=1)
