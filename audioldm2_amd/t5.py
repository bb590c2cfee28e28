"""FLAN-T5 text conditioner on the MI355X.

Drop-in for `audioldm2.latent_diffusion.modules.encoders.modules.FlanT5HiddenState` (encoders/modules.py:113-198): same
constructor (`text_encoder_name`, `freeze_text_encoder`), same `forward(batch_of_strings) -> [hidden [B, T, 1024], mask [B, T]]`
and `get_unconditional_condition(batchsize)`; `self.model` holds parameters under `transformers.T5EncoderModel`'s state-dict
keys (`shared.weight`, `encoder.block.i.layer.0.SelfAttention.{q,k,v,o}.weight`, `...relative_attention_bias.weight`,
`...layer_norm.weight`, `encoder.block.i.layer.1.DenseReluDense.{wi_0,wi_1,wo}.weight`, `encoder.final_layer_norm.weight`), so a
checkpoint's `cond_stage_models.*.model.*` entries load unchanged.

The tokenizer is the reference's (`AutoTokenizer.from_pretrained`, a sentencepiece model from the Hub); where the Hub is out
of reach the module still builds (flan-t5-large's published geometry) and `encode_tokens(input_ids, attention_mask)` is the
entry point.  Everything after the tokenizer runs on the kernel library: T5LayerNorm = aldm_rmsnorm, q/k/v and the gated
FF's wi_0/wi_1 as fused GEMMs (the tanh-GELU gate in the GEMM epilogue), un-scaled attention as Q K^T (batched NT GEMM) ->
aldm_softmax_rows_bias (bucketed relative-position bias + padding mask) -> P V.  torch gathers embedding rows and builds the
bucket table on the host once per sequence length.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops

# device of the packed weights and activations; tests/test_host_logic.py points it at the CPU next to torch stand-ins for `ops`
_DEV = torch.device("cuda")

FLAN_T5_LARGE = dict(vocab_size=32128, d_model=1024, d_kv=64, d_ff=2816, num_layers=24, num_heads=16,
                     relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6)


class _W(nn.Module):
    """A bias-free Linear / T5LayerNorm weight holder with transformers' attribute name `weight`."""

    def __init__(self, *shape):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(*shape) * (0.02 if len(shape) > 1 else 0.0) + (1.0 if len(shape) == 1 else 0.0))


class _SelfAttention(nn.Module):
    def __init__(self, cfg, has_bias_table):
        super().__init__()
        inner = cfg["num_heads"] * cfg["d_kv"]
        self.q, self.k, self.v = _W(inner, cfg["d_model"]), _W(inner, cfg["d_model"]), _W(inner, cfg["d_model"])
        self.o = _W(cfg["d_model"], inner)
        if has_bias_table:
            self.relative_attention_bias = nn.Embedding(cfg["relative_attention_num_buckets"], cfg["num_heads"])


class _LayerSA(nn.Module):
    def __init__(self, cfg, first):
        super().__init__()
        self.SelfAttention = _SelfAttention(cfg, first)
        self.layer_norm = _W(cfg["d_model"])


class _Dense(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.wi_0, self.wi_1 = _W(cfg["d_ff"], cfg["d_model"]), _W(cfg["d_ff"], cfg["d_model"])
        self.wo = _W(cfg["d_model"], cfg["d_ff"])


class _LayerFF(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.DenseReluDense = _Dense(cfg)
        self.layer_norm = _W(cfg["d_model"])


class _Block(nn.Module):
    def __init__(self, cfg, first):
        super().__init__()
        self.layer = nn.ModuleList([_LayerSA(cfg, first), _LayerFF(cfg)])


class _Stack(nn.Module):
    def __init__(self, cfg, shared):
        super().__init__()
        self.embed_tokens = shared
        self.block = nn.ModuleList([_Block(cfg, i == 0) for i in range(cfg["num_layers"])])
        self.final_layer_norm = _W(cfg["d_model"])


class T5EncoderModel(nn.Module):
    """Parameter holder with transformers.T5EncoderModel's names; forward on the HIP ops."""

    def __init__(self, cfg: dict):
        super().__init__()
        self.cfg = dict(cfg)
        self.shared = nn.Embedding(cfg["vocab_size"], cfg["d_model"])
        self.encoder = _Stack(cfg, self.shared)   # embed_tokens is tied to shared (both keys appear in the state dict)
        self._pk = None
        self._bias_cache = {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    def invalidate_packed(self):
        self._pk = None
        self._bias_cache = {}

    def _prepare(self):
        if self._pk is None:
            f = lambda t: t.detach().float().to(_DEV).contiguous()
            layers = []
            for b in self.encoder.block:
                sa, ff = b.layer[0].SelfAttention, b.layer[1].DenseReluDense
                layers.append(dict(
                    ln1=f(b.layer[0].layer_norm.weight), ln2=f(b.layer[1].layer_norm.weight),
                    qkv=ops.pack_conv(torch.cat([sa.q.weight, sa.k.weight, sa.v.weight], 0)),
                    o=ops.pack_conv(sa.o.weight),
                    # gated FF as ONE GEMM: value = wi_1 x, gate = wi_0 x, out = value * gelu_new(gate) in the epilogue
                    wi=ops.pack_geglu(torch.cat([ff.wi_1.weight, ff.wi_0.weight], 0), None),
                    wo=ops.pack_conv(ff.wo.weight)))
            self._pk = dict(layers=layers, table=f(self.shared.weight), lnf=f(self.encoder.final_layer_norm.weight),
                            rel=f(self.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight))
        return self._pk

    def _position_bias(self, T: int) -> torch.Tensor:
        """[H, T, T]: layer 0's table gathered at the bidirectional buckets of key - query (T5Attention.compute_bias)."""
        if T not in self._bias_cache:
            cfg, pk = self.cfg, self._prepare()
            nb = cfg["relative_attention_num_buckets"] // 2
            pos = torch.arange(T)
            rel = pos[None, :] - pos[:, None]
            out = (rel > 0).long() * nb
            rel = rel.abs()
            max_exact = nb // 2
            large = max_exact + (torch.log(rel.float() / max_exact) /
                                 math.log(cfg["relative_attention_max_distance"] / max_exact) * (nb - max_exact)).long()
            large = torch.min(large, torch.full_like(large, nb - 1))
            buckets = out + torch.where(rel < max_exact, rel, large)
            self._bias_cache[T] = pk["rel"][buckets.to(pk["rel"].device)].permute(2, 0, 1).contiguous()
        return self._bias_cache[T]

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None):
        """-> (last_hidden_state [B, T, d_model],) like transformers' model(...)[0] consumers expect."""
        pk, cfg = self._prepare(), self.cfg
        dev = pk["table"].device
        ids = input_ids.to(dev)
        B, T = ids.shape
        keymask = (torch.ones(B, T, device=dev) if attention_mask is None else attention_mask.to(dev)).float().contiguous()
        H, dk, eps = cfg["num_heads"], cfg["d_kv"], cfg["layer_norm_epsilon"]
        Tp = (T + 3) // 4 * 4   # the batched GEMMs want K % 4 == 0: keys are padded (and masked) up to Tp
        bias = self._position_bias(T)
        if Tp != T:
            bias = torch.nn.functional.pad(bias, (0, Tp - T)).contiguous()
            keymask_p = torch.nn.functional.pad(keymask, (0, Tp - T)).contiguous()
        else:
            keymask_p = keymask
        x = pk["table"].index_select(0, ids.reshape(-1)).view(B * T, -1)
        Z = B * H
        for L in pk["layers"]:
            n = ops.rmsnorm(x, L["ln1"], eps)
            qkv = ops.linear(n, L["qkv"]).view(B, T, 3, H, dk)
            q = qkv[:, :, 0].permute(0, 2, 1, 3).reshape(Z, T, dk)                       # head split: copies
            k = torch.zeros((Z, Tp, dk), device=dev)
            v = torch.zeros((Z, Tp, dk), device=dev)
            k.view(B, H, Tp, dk)[:, :, :T] = qkv[:, :, 1].permute(0, 2, 1, 3)
            v.view(B, H, Tp, dk)[:, :, :T] = qkv[:, :, 2].permute(0, 2, 1, 3)
            s = ops.gemm_nt(q, k)                                                        # [Z, T, Tp], no 1/sqrt(d) in T5
            p = ops.softmax_rows_bias(s.view(B, H, T, Tp), bias, keymask_p)
            o = ops.gemm_packed_batched(p.view(Z, T, Tp), ops.pack_kn(v), Tp, dk)        # [Z, T, dk]
            o = o.view(B, H, T, dk).permute(0, 2, 1, 3).reshape(B * T, H * dk)           # head merge: copy
            x = ops.linear(o, L["o"], res=x)
            n = ops.rmsnorm(x, L["ln2"], eps)
            g = ops.linear_geglu(n, L["wi"], gate_act=ops.ACT_GELU_TANH)
            x = ops.linear(g, L["wo"], res=x)
        return (ops.rmsnorm(x, pk["lnf"], eps).view(B, T, -1),)


class FlanT5HiddenState(nn.Module):
    def __init__(self, text_encoder_name="google/flan-t5-large", freeze_text_encoder=True, config: dict = None):
        super().__init__()
        self.freeze_text_encoder = freeze_text_encoder
        self.tokenizer = None
        cfg = config
        try:  # the reference's two Hub calls (encoders/modules.py:126-127); offline both are unreachable
            from transformers import AutoTokenizer, T5Config
            self.tokenizer = AutoTokenizer.from_pretrained(text_encoder_name)
            if cfg is None:
                c = T5Config.from_pretrained(text_encoder_name)
                cfg = {k: getattr(c, k) for k in FLAN_T5_LARGE}
        except Exception:
            pass
        if cfg is None:
            if "flan-t5-large" not in text_encoder_name:
                raise RuntimeError(f"no configuration for {text_encoder_name!r} without Hub access; pass config=")
            cfg = dict(FLAN_T5_LARGE)
        self.model = T5EncoderModel(cfg)
        for p in self.model.parameters():
            p.requires_grad = False
        self.empty_hidden_state_cfg = None
        self.device = None

    def get_unconditional_condition(self, batchsize):
        """encoders/modules.py:138-154: the encoding of "" tiled, mask all ones."""
        if self.empty_hidden_state_cfg is None:
            self.empty_hidden_state_cfg, _ = self([""])
        hidden_state = torch.cat([self.empty_hidden_state_cfg] * batchsize).float()
        attention_mask = torch.ones((batchsize, hidden_state.size(1)), device=hidden_state.device).float()
        return [hidden_state, attention_mask]

    def forward(self, batch):
        return self.encode_text(batch)

    def encode_text(self, prompt):
        """encoders/modules.py:173-198"""
        if self.tokenizer is None:
            raise RuntimeError("FlanT5HiddenState: the tokenizer could not be loaded (no Hub access); tokenize elsewhere "
                               "and call encode_tokens(input_ids, attention_mask)")
        batch = self.tokenizer(prompt, max_length=128, padding=True, truncation=True, return_tensors="pt")
        return self.encode_tokens(batch.input_ids, batch.attention_mask)

    def encode_tokens(self, input_ids, attention_mask):
        hidden = self.model(input_ids=input_ids, attention_mask=attention_mask)[0]
        return [hidden.detach(), attention_mask.to(hidden.device).float()]
