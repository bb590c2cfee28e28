"""oracle/clap_text.py — CPU restatement of the CLAP TEXT tower (second half).  TEST INFRASTRUCTURE ONLY.

Reference: `CLAPAudioEmbeddingClassifierFreev2.forward` in "text" mode (audioldm2/latent_diffusion/modules/encoders/modules.py:
717-735) -> `CLAP.get_text_embedding` (audioldm2/clap/open_clip/model.py:730-747) -> `encode_text` roberta branch (:656-663):
`RobertaModel(RobertaConfig.from_pretrained("roberta-base"))(input_ids, attention_mask)["pooler_output"]` -> `text_projection`
(Linear 768->512, ReLU, Linear 512->512; :525-529) -> F.normalize.  Tokens come from `RobertaTokenizer` at padding="max_length",
max_length 512 (modules.py:737-745); the tokenizer needs the Hub, so the boundary here is token ids.
Third-party arithmetic: `transformers==4.30.2` RoBERTa — not vendored; restated from its published algorithm (BERT post-LN
encoder, erf GELU, learned absolute positions offset by padding_idx, tanh pooler over the first token) and pinned by a
fixture generated from the installed transformers' RobertaModel + the reference's projection head, built exactly as
model.py:513-529 builds them (tests/golden/clap_text_*.npz, oracle/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

# roberta-base (config.json of the checkpoint model.py:515 names)
ROBERTA_BASE = dict(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                    max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1)
JOINT_DIM = 512


def roberta_pooled(sd: Dict[str, torch.Tensor], cfg: dict, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                   prefix: str = "text_branch.") -> torch.Tensor:
    """RobertaModel(input_ids, attention_mask)["pooler_output"] in eval mode: [B, hidden]."""
    H, eps, pad = cfg["num_attention_heads"], cfg["layer_norm_eps"], cfg["pad_token_id"]
    B, T = input_ids.shape
    m = (input_ids != pad).long()
    pos = torch.cumsum(m, dim=1) * m + pad                      # create_position_ids_from_input_ids
    e = prefix + "embeddings."
    x = F.embedding(input_ids, sd[e + "word_embeddings.weight"]) + sd[e + "token_type_embeddings.weight"][0] + \
        F.embedding(pos, sd[e + "position_embeddings.weight"])
    x = F.layer_norm(x, (x.shape[-1],), sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], eps)
    add = (1.0 - attention_mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    d = x.shape[-1] // H
    for l in range(cfg["num_hidden_layers"]):
        p = f"{prefix}encoder.layer.{l}."
        lin = lambda t, n: F.linear(t, sd[p + n + ".weight"], sd[p + n + ".bias"])
        hs = lambda t: t.view(B, T, H, d).transpose(1, 2)
        q, k, v = hs(lin(x, "attention.self.query")), hs(lin(x, "attention.self.key")), hs(lin(x, "attention.self.value"))
        a = F.softmax(q @ k.transpose(-1, -2) / math.sqrt(d) + add, dim=-1) @ v
        a = lin(a.transpose(1, 2).reshape(B, T, H * d), "attention.output.dense")
        x = F.layer_norm(a + x, (x.shape[-1],), sd[p + "attention.output.LayerNorm.weight"],
                         sd[p + "attention.output.LayerNorm.bias"], eps)
        h = F.gelu(lin(x, "intermediate.dense"))
        h = lin(h, "output.dense")
        x = F.layer_norm(h + x, (x.shape[-1],), sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    return torch.tanh(F.linear(x[:, 0], sd[prefix + "pooler.dense.weight"], sd[prefix + "pooler.dense.bias"]))


def text_embedding(sd: Dict[str, torch.Tensor], cfg: dict, input_ids: torch.Tensor, attention_mask: torch.Tensor
                   ) -> torch.Tensor:
    """CLAP.get_text_embedding (model.py:730-747): pooled -> text_projection (:525-529) -> L2 normalise.  [B, 512]."""
    x = roberta_pooled(sd, cfg, input_ids, attention_mask)
    x = F.linear(x, sd["text_projection.0.weight"], sd["text_projection.0.bias"])
    x = F.linear(torch.relu(x), sd["text_projection.2.weight"], sd["text_projection.2.bias"])
    return F.normalize(x, dim=-1)
