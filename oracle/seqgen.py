"""oracle/seqgen.py — CPU restatement of the AudioMAE-token sequence generator (the next row after
the sampling path).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference: `Sequence2AudioMAE.generate` (audioldm2/audiomae_gen/sequence_input.py:294-325) over
`get_input_sequence_and_mask` (:136-199) and `add_sos_eos_tokens` (:109-124); the network is
`transformers.GPT2Model(GPT2Config.from_pretrained("gpt2"))` (:69) driven with `inputs_embeds` / `attention_mask`.
Third-party arithmetic: `transformers==4.30.2` (requirements pin) `GPT2Model` — not vendored by the reference; restated
here from its published algorithm (pre-LN blocks, `Conv1D` = x @ W + b with W stored [in, out], scaled dot-product attention
with causal + additive padding mask at finfo.min, `gelu_new`, learned absolute positions) and pinned by a fixture generated
from the REAL reference class running on the installed transformers (tests/golden/seqgen_*.npz, oracle/make_golden.py).

Two evaluation orders of the same function:
  generate_full    what the reference does: every step re-runs GPT-2 over the whole prefix (O(n^2) forwards);
  generate_cached  keys / values of earlier positions are kept (what an accelerated decode does); identical up to fp32
                   summation order because a causal model's activations at position i do not depend on later positions.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

N_LAYER, N_HEAD, N_EMBD, LN_EPS = 12, 12, 768, 1e-5  # GPT2Config defaults == the "gpt2" checkpoint's config


def gelu_new(x: torch.Tensor) -> torch.Tensor:
    """transformers' NewGELUActivation (GPT-2's `activation_function="gelu_new"`)."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def input_sequence_and_mask(sd: Dict[str, torch.Tensor], cond: Dict[str, object], keys: Sequence[str],
                            mae_token_num: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """sequence_input.py:136-199: every conditioning sequence is projected to 768 (`input_sequence_embed_linear[i]`),
    wrapped in its own learned start / end tokens (row i of the two 32-entry tables, :109-124), concatenated along time
    and truncated to 1024 - mae_token_num positions.  A tensor-valued cond ([B, T, D]) gets an all-ones mask."""
    embeds, masks = [], []
    for i, key in enumerate(keys):
        c = cond[key]
        if isinstance(c, (list, tuple)):
            x, m = c
        else:
            x, m = c, torch.ones(c.shape[0], c.shape[1])
        x = F.linear(x, sd[f"input_sequence_embed_linear.{i}.weight"], sd[f"input_sequence_embed_linear.{i}.bias"])
        B = x.shape[0]
        sos = sd["start_of_sequence_tokens.weight"][i].expand(B, 1, -1)
        eos = sd["end_of_sequence_tokens.weight"][i].expand(B, 1, -1)
        one = torch.ones(B, 1)
        embeds.append(torch.cat([sos, x, eos], dim=1))
        masks.append(torch.cat([one, m.float(), one], dim=1))
    x, m = torch.cat(embeds, dim=1), torch.cat(masks, dim=1)
    max_len = 1024 - mae_token_num
    return x[:, :max_len], m[:, :max_len]


def _block(sd, l: int, h: torch.Tensor, k_prev, v_prev, add_mask: torch.Tensor):
    """One GPT-2 block on the positions in `h` ([B, T, 768]) given the keys / values of the earlier positions
    (k_prev / v_prev [B, heads, P, 64] or None).  add_mask [B, 1, 1, P + T]: 0 or finfo.min per key position."""
    p = f"model.h.{l}."
    B, T, _ = h.shape
    a = F.layer_norm(h, (N_EMBD,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], LN_EPS)
    qkv = a @ sd[p + "attn.c_attn.weight"] + sd[p + "attn.c_attn.bias"]
    q, k, v = (t.view(B, T, N_HEAD, 64).transpose(1, 2) for t in qkv.split(N_EMBD, dim=2))
    if k_prev is not None:
        k, v = torch.cat([k_prev, k], dim=2), torch.cat([v_prev, v], dim=2)
    P = k.shape[2] - T
    s = (q @ k.transpose(-1, -2)) / math.sqrt(64.0)
    fmin = torch.finfo(s.dtype).min
    causal = torch.ones(T, P + T, dtype=torch.bool).tril(diagonal=P)  # query i (absolute P + i) sees keys <= P + i
    s = torch.where(causal, s, torch.full([], fmin)) + add_mask
    o = (s.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, T, N_EMBD)
    h = h + (o @ sd[p + "attn.c_proj.weight"] + sd[p + "attn.c_proj.bias"])
    m = F.layer_norm(h, (N_EMBD,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], LN_EPS)
    m = gelu_new(m @ sd[p + "mlp.c_fc.weight"] + sd[p + "mlp.c_fc.bias"])
    return h + (m @ sd[p + "mlp.c_proj.weight"] + sd[p + "mlp.c_proj.bias"]), k, v


def gpt2_forward(sd, x: torch.Tensor, mask: torch.Tensor, cache: List = None, pos0: int = 0):
    """GPT2Model(inputs_embeds=x, attention_mask=mask)["last_hidden_state"] for positions pos0 .. pos0 + T - 1; `mask`
    covers every key position 0 .. pos0 + T - 1.  `cache` (list of (k, v) per layer, or None) holds earlier positions."""
    B, T, _ = x.shape
    h = x + sd["model.wpe.weight"][pos0:pos0 + T]
    add = ((1.0 - mask.float()) * torch.finfo(torch.float32).min)[:, None, None, :]
    new_cache = []
    for l in range(N_LAYER):
        kp, vp = cache[l] if cache is not None else (None, None)
        h, k, v = _block(sd, l, h, kp, vp, add)
        new_cache.append((k, v))
    return F.layer_norm(h, (N_EMBD,), sd["model.ln_f.weight"], sd["model.ln_f.bias"], LN_EPS), new_cache


def generate_full(sd, x: torch.Tensor, mask: torch.Tensor, steps: int) -> torch.Tensor:
    """sequence_input.py:308-323 as written: `steps` full forwards, the last hidden state becomes the next input."""
    n0 = x.shape[1]
    for _ in range(steps):
        out, _ = gpt2_forward(sd, x, mask)
        x = torch.cat([x, out[:, -1:, :]], dim=1)
        mask = torch.cat([mask, torch.ones(mask.shape[0], 1)], dim=1)
    return x[:, n0:]


def generate_cached(sd, x: torch.Tensor, mask: torch.Tensor, steps: int) -> torch.Tensor:
    """Same function with a key / value cache: one forward over the conditioning prefix, then one position per step."""
    out, cache = gpt2_forward(sd, x, mask)
    tok, outs, pos = out[:, -1:, :], [], x.shape[1]
    for _ in range(steps):
        outs.append(tok)
        if len(outs) == steps:
            break
        mask = torch.cat([mask, torch.ones(mask.shape[0], 1)], dim=1)
        out, cache = gpt2_forward(sd, tok, mask, cache, pos)
        tok, pos = out, pos + 1
    return torch.cat(outs, dim=1)
