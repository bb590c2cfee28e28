"""oracle/t5.py — CPU restatement of the FLAN-T5 text conditioner.  TEST INFRASTRUCTURE ONLY.

Reference: `FlanT5HiddenState.encode_text` (audioldm2/latent_diffusion/modules/encoders/modules.py:173-198): tokenizer ->
`transformers.T5EncoderModel(T5Config.from_pretrained("google/flan-t5-large"))(input_ids, attention_mask)[0]`, returned
with the float attention mask; `get_unconditional_condition` (:138-154) = the encoding of "" tiled, mask all ones.
Third-party arithmetic: `transformers==4.30.2` (requirements pin) T5 encoder — not vendored by the reference; restated here
from its published algorithm (T5 v1.1 / FLAN: RMS "T5LayerNorm" eps 1e-6, pre-norm blocks, un-scaled dot-product attention
with a learned bucketed relative-position bias shared by all layers, additive finfo.min padding mask, gated-GELU FF with
tanh GELU) and pinned by a fixture generated from the REAL reference class running on the installed transformers with the
Hub calls (tokenizer, config) replaced by local equivalents (tests/golden/t5_*.npz, oracle/make_golden.py).
The tokenizer (sentencepiece model from the Hub) is out of reach offline: the boundary is token ids.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

# google/flan-t5-large (config.json of the checkpoint the reference names, modules.py:122)
FLAN_T5_LARGE = dict(vocab_size=32128, d_model=1024, d_kv=64, d_ff=2816, num_layers=24, num_heads=16,
                     relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6,
                     feed_forward_proj="gated-gelu", tie_word_embeddings=False)


def gelu_new(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """T5LayerNorm: no mean subtraction, no bias; variance in fp32."""
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def relative_position_bucket(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """Bidirectional bucketing of rel = key_position - query_position (T5Attention._relative_position_bucket)."""
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(is_small, rel, large)


def position_bias(sd: Dict[str, torch.Tensor], T: int, cfg: dict) -> torch.Tensor:
    """[H, T, T] bias of layer 0's relative_attention_bias table, reused by every layer."""
    pos = torch.arange(T)
    buckets = relative_position_bucket(pos[None, :] - pos[:, None], cfg["relative_attention_num_buckets"],
                                       cfg["relative_attention_max_distance"])
    w = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]   # [buckets, H]
    return w[buckets].permute(2, 0, 1).contiguous()


def encoder_forward(sd: Dict[str, torch.Tensor], cfg: dict, input_ids: torch.Tensor, attention_mask: torch.Tensor
                    ) -> torch.Tensor:
    """T5EncoderModel(input_ids, attention_mask)[0] in eval mode: [B, T, d_model]."""
    H, dk, eps = cfg["num_heads"], cfg["d_kv"], cfg["layer_norm_epsilon"]
    B, T = input_ids.shape
    x = F.embedding(input_ids, sd["shared.weight"])
    bias = position_bias(sd, T, cfg)[None] + (1.0 - attention_mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    for l in range(cfg["num_layers"]):
        p = f"encoder.block.{l}.layer."
        n = rms_norm(x, sd[p + "0.layer_norm.weight"], eps)
        hs = lambda t: t.view(B, T, H, dk).transpose(1, 2)
        q, k, v = (hs(F.linear(n, sd[p + f"0.SelfAttention.{c}.weight"])) for c in "qkv")
        scores = q @ k.transpose(-1, -2) + bias                                  # no 1/sqrt(d): folded into the init
        a = F.softmax(scores.float(), dim=-1) @ v
        x = x + F.linear(a.transpose(1, 2).reshape(B, T, H * dk), sd[p + "0.SelfAttention.o.weight"])
        n = rms_norm(x, sd[p + "1.layer_norm.weight"], eps)
        g = gelu_new(F.linear(n, sd[p + "1.DenseReluDense.wi_0.weight"])) * F.linear(n, sd[p + "1.DenseReluDense.wi_1.weight"])
        x = x + F.linear(g, sd[p + "1.DenseReluDense.wo.weight"])
    return rms_norm(x, sd["encoder.final_layer_norm.weight"], eps)


def encode_tokens(sd, cfg, input_ids, attention_mask) -> Tuple[torch.Tensor, torch.Tensor]:
    """FlanT5HiddenState.encode_text after the tokenizer (modules.py:182-198): [hidden states, float mask]."""
    return encoder_forward(sd, cfg, input_ids, attention_mask), attention_mask.float()
