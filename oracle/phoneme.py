"""oracle/phoneme.py — CPU restatement of the VITS phoneme encoder, the second half of SURVEY.md §8(f) rank 1 (config 5, TTS:
`crossattn_vits_phoneme`).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: `PhonemeEncoder.forward` (audioldm2/latent_diffusion/modules/encoders/modules.py:94-110) over `TextEncoder.forward`
(phoneme_encoder/encoder.py:39-50) and `attentions.Encoder` (phoneme_encoder/attentions.py:26-87): 6 post-LN layers of
windowed relative-position self-attention (`MultiHeadAttention`, :183-330, window_size 4, 2 heads x 96) and a k=3 conv FFN
(`FFN`, :374-430, relu).  Functional torch fp32, each step citing the line it follows; pinned by a fixture generated from the
REAL reference classes (tests/golden/phoneme_*.npz, oracle/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

HIDDEN, FILTER, HEADS, LAYERS, KSIZE, WINDOW, LN_EPS = 192, 768, 2, 6, 3, 4, 1e-5  # modules.py:46-55, attentions.py:35


def rel_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, emb_k: torch.Tensor, emb_v: torch.Tensor,
                  mask: torch.Tensor) -> torch.Tensor:
    """attentions.py:239-289 for self-attention with window_size = 4, heads_share (emb_* [1, 9, d]).  q/k/v [B, H, T, d];
    mask [B, T] (1 = real token): the reference's attn_mask is mask_i * mask_j (:76), scores at mask == 0 are SET to -1e4
    (:263).  The pad/reshape skews of :317-361 are index shifts: rel_logits[i, j - i + 4] lands on score (i, j) for
    |j - i| <= 4 (zero elsewhere), and the value term reads p[i, j] at the same offsets."""
    B, H, T, d = q.shape
    qs = q / math.sqrt(d)                                            # :247
    scores = qs @ k.transpose(-2, -1)                                # :247
    rel_logits = qs @ emb_k[0].t()                                   # :252-255 -> [B, H, T, 9]
    idx = torch.arange(T)
    off = idx[None, :] - idx[:, None]                                # j - i
    inside = off.abs() <= WINDOW
    gather = (off.clamp(-WINDOW, WINDOW) + WINDOW)                   # [T, T] in 0..8
    local = torch.gather(rel_logits, 3, gather[None, None].expand(B, H, T, T))
    scores = scores + torch.where(inside[None, None], local, torch.zeros(()))   # :256-257
    am = mask[:, None, :, None] * mask[:, None, None, :]             # :76
    scores = scores.masked_fill(am == 0, -1e4)                       # :263
    p = F.softmax(scores, dim=-1)                                    # :274
    out = p @ v                                                      # :276
    # :277-283: relative_weights[i, r] = p[i, i + r - 4] (0 outside the sequence), times emb_v[r]
    pw = torch.where(inside[None, None], p, torch.zeros(()))
    rw = torch.zeros(B, H, T, 2 * WINDOW + 1)
    rw.scatter_add_(3, gather[None, None].expand(B, H, T, T), pw)
    return out + rw @ emb_v[0]


def layer_norm_c(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """attentions.py:11-23: LayerNorm over channels of [B, C, T] == over the last dim of the channels-last view."""
    return F.layer_norm(x, (x.shape[-1],), gamma, beta, LN_EPS)


def encoder_forward(sd: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """attentions.Encoder.forward (:75-87) on channels-last x [B, T, 192]; mask [B, T]."""
    m3 = mask[:, :, None]
    x = x * m3                                                       # :77
    for i in range(LAYERS):
        a = f"{prefix}attn_layers.{i}."
        lin = lambda t, n: F.linear(t, sd[a + n + ".weight"][:, :, 0], sd[a + n + ".bias"])   # Conv1d k=1 (:202-205)
        B, T, C = x.shape
        hs = lambda t: t.view(B, T, HEADS, C // HEADS).transpose(1, 2)                         # :241-244
        y = rel_attention(hs(lin(x, "conv_q")), hs(lin(x, "conv_k")), hs(lin(x, "conv_v")), sd[a + "emb_rel_k"],
                          sd[a + "emb_rel_v"], mask)
        y = lin(y.transpose(1, 2).reshape(B, T, C), "conv_o")                                  # :286-288, :235
        x = layer_norm_c(x + y, sd[f"{prefix}norm_layers_1.{i}.gamma"], sd[f"{prefix}norm_layers_1.{i}.beta"])  # :79-81
        f = f"{prefix}ffn_layers.{i}."
        conv = lambda t, n: F.conv1d(F.pad(t.transpose(1, 2), (1, 1)), sd[f + n + ".weight"], sd[f + n + ".bias"]).transpose(1, 2)
        y = conv(x * m3, "conv_1")                                                             # :406, :421-430 same padding
        y = torch.relu(y)                                                                      # :410
        y = conv(y * m3, "conv_2") * m3                                                        # :412-413
        x = layer_norm_c(x + y, sd[f"{prefix}norm_layers_2.{i}.gamma"], sd[f"{prefix}norm_layers_2.{i}.beta"])  # :83-85
    return x * m3                                                    # :86


def phoneme_encoder_forward(sd: Dict[str, torch.Tensor], phoneme_idx: torch.Tensor, pad_token_id: int
                            ) -> Tuple[torch.Tensor, torch.Tensor]:
    """PhonemeEncoder.forward (modules.py:94-110): returns [text_emb [B, T, 192], mask [B, T]]."""
    length = (phoneme_idx != pad_token_id).sum(-1)                   # modules.py:75-82
    T = phoneme_idx.shape[1]
    x = F.embedding(phoneme_idx, sd["text_encoder.emb.weight"]) * math.sqrt(HIDDEN)           # encoder.py:40
    mask = (torch.arange(T)[None, :] < length[:, None]).float()      # commons.sequence_mask, encoder.py:42-44
    x = encoder_forward(sd, "text_encoder.encoder.", x * mask[:, :, None], mask)              # encoder.py:46
    x = x + sd["learnable_positional_embedding"][0].t()[None]        # modules.py:103 ([1, 192, T] added to [B, 192, T])
    return x, mask


def unconditional(sd: Dict[str, torch.Tensor], batchsize: int, pad_length: int, pad_token_id: int):
    """PhonemeEncoder.get_unconditional_condition (modules.py:63-67): the encoder on an all-pad sequence."""
    return phoneme_encoder_forward(sd, torch.full((batchsize, pad_length), pad_token_id, dtype=torch.long), pad_token_id)
