"""CLAP text tower on the MI355X (second half) behind the reference's conditioner class.

`CLAPAudioEmbeddingClassifierFreev2` mirrors `audioldm2.latent_diffusion.modules.encoders.modules.CLAPAudioEmbeddingClassifierFreev2`
(encoders/modules.py:546-745): same constructor keywords, `forward(batch)` in `embed_mode="text"` (the mode every AudioLDM2
config uses for the FiLM / sequence conditioning, utils.py:146-155), `get_unconditional_condition`, and the
`unconditional_prob` draw of modules.py:731-733 (a `torch.rand(1)` per batch element from the host generator — RNG contract R2).
`self.model` holds the text side of `clap.open_clip.model.CLAP` under its state-dict keys (`text_branch.*` = transformers
RobertaModel, `text_projection.{0,2}.*`), so `cond_stage_models.*.model.text_*` / `clap.model.text_*` checkpoint entries load
(strict=False skips the audio tower's keys).

embed_mode="audio" (CLAP re-ranking of n_candidate_gen_per_text > 1, ddpm.py:1554-1568) needs the HTSAT audio tower, which is
NOT built here: it raises NotImplementedError — at construction of the re-ranker, not after sampling.

The tokenizer is the reference's `RobertaTokenizer.from_pretrained("roberta-base")`; where the Hub is out of reach the module
still builds (roberta-base's published geometry) and `encode_tokens(input_ids, attention_mask)` is the entry point.
Compute: embedding gathers by torch, everything else through the C ABI (aldm_layernorm, aldm_igemm with fused bias / erf-GELU /
tanh / ReLU / residual epilogues, Q K^T and P V as batched GEMMs around aldm_softmax_rows_masked, aldm_row_l2norm +
aldm_rowscale_add for F.normalize).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops

ROBERTA_BASE = dict(vocab_size=50265, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                    max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, pad_token_id=1)
JOINT_DIM = 512


class _Emb(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        h = cfg["hidden_size"]
        self.word_embeddings = nn.Embedding(cfg["vocab_size"], h)
        self.position_embeddings = nn.Embedding(cfg["max_position_embeddings"], h)
        self.token_type_embeddings = nn.Embedding(cfg["type_vocab_size"], h)
        self.LayerNorm = nn.LayerNorm(h)


class _SelfAttn(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(h, h), nn.Linear(h, h), nn.Linear(h, h)


class _Out(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.dense = nn.Linear(i, o)
        self.LayerNorm = nn.LayerNorm(o)


class _Attn(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.self = _SelfAttn(h)
        self.output = _Out(h, h)


class _Inter(nn.Module):
    def __init__(self, h, i):
        super().__init__()
        self.dense = nn.Linear(h, i)


class _Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        h, i = cfg["hidden_size"], cfg["intermediate_size"]
        self.attention = _Attn(h)
        self.intermediate = _Inter(h, i)
        self.output = _Out(i, h)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(cfg) for _ in range(cfg["num_hidden_layers"])])


class _Pooler(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.dense = nn.Linear(h, h)


class _Roberta(nn.Module):
    """Parameter holder with transformers.RobertaModel's names."""

    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Emb(cfg)
        self.encoder = _Encoder(cfg)
        self.pooler = _Pooler(cfg["hidden_size"])


class CLAPTextModel(nn.Module):
    """The text side of clap.open_clip.model.CLAP (model.py:513-529, 656-663, 730-747)."""

    def __init__(self, cfg: dict = None):
        super().__init__()
        self.cfg = dict(cfg or ROBERTA_BASE)
        self.text_branch = _Roberta(self.cfg)
        self.text_projection = nn.Sequential(nn.Linear(self.cfg["hidden_size"], JOINT_DIM), nn.ReLU(),
                                             nn.Linear(JOINT_DIM, JOINT_DIM))
        self._pk = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    def invalidate_packed(self):
        self._pk = None

    def _prepare(self):
        if self._pk is None:
            f = lambda t: t.detach().float().cuda().contiguous()
            tb = self.text_branch
            layers = []
            for L in tb.encoder.layer:
                a = L.attention
                layers.append(dict(
                    qkv=ops.pack_conv(torch.cat([a.self.query.weight, a.self.key.weight, a.self.value.weight], 0),
                                      torch.cat([a.self.query.bias, a.self.key.bias, a.self.value.bias], 0)),
                    ao=ops.pack_conv(a.output.dense.weight, a.output.dense.bias),
                    ln1=(f(a.output.LayerNorm.weight), f(a.output.LayerNorm.bias)),
                    fi=ops.pack_conv(L.intermediate.dense.weight, L.intermediate.dense.bias),
                    fo=ops.pack_conv(L.output.dense.weight, L.output.dense.bias),
                    ln2=(f(L.output.LayerNorm.weight), f(L.output.LayerNorm.bias))))
            e = tb.embeddings
            self._pk = dict(layers=layers, word=f(e.word_embeddings.weight), pos=f(e.position_embeddings.weight),
                            tok0=f(e.token_type_embeddings.weight[0]), eln=(f(e.LayerNorm.weight), f(e.LayerNorm.bias)),
                            pool=ops.pack_conv(tb.pooler.dense.weight, tb.pooler.dense.bias),
                            p0=ops.pack_conv(self.text_projection[0].weight, self.text_projection[0].bias),
                            p2=ops.pack_conv(self.text_projection[2].weight, self.text_projection[2].bias))
        return self._pk

    @torch.no_grad()
    def get_text_embedding(self, data) -> torch.Tensor:
        """model.py:730-747: {"input_ids", "attention_mask"} [B, T] -> L2-normalised [B, 512]."""
        pk, cfg = self._prepare(), self.cfg
        dev = pk["word"].device
        ids = data["input_ids"].to(dev)
        am = data["attention_mask"].to(dev)
        B, T = ids.shape
        assert T % 4 == 0, "pad the token batch to a multiple of 4 (the reference pads to max_length = 512)"
        H = cfg["num_attention_heads"]
        C = cfg["hidden_size"]
        d = C // H
        m = (ids != cfg["pad_token_id"]).long()
        pos = torch.cumsum(m, dim=1) * m + cfg["pad_token_id"]      # create_position_ids_from_input_ids (host-size index math)
        x = pk["word"].index_select(0, ids.reshape(-1))
        x = ops.axpby(x, pk["pos"].index_select(0, pos.reshape(-1)), 1.0, 1.0)
        x = ops.axpby(x, pk["tok0"].expand(B * T, C).contiguous(), 1.0, 1.0)
        x = ops.layernorm(x, *pk["eln"], eps=cfg["layer_norm_eps"])
        keymask = am.float().contiguous()
        Z = B * H
        for L in pk["layers"]:
            qkv = ops.linear(x, L["qkv"]).view(B, T, 3, H, d)
            q = qkv[:, :, 0].permute(0, 2, 1, 3).reshape(Z, T, d)
            k = qkv[:, :, 1].permute(0, 2, 1, 3).reshape(Z, T, d)
            v = qkv[:, :, 2].permute(0, 2, 1, 3).reshape(Z, T, d)
            s = ops.gemm_nt(q, k, alpha=1.0 / math.sqrt(d))
            p = ops.softmax_rows_masked(s.view(B, H, T, T), keymask, T)      # q_pos0 = T: no causal limit
            o = ops.gemm_packed_batched(p.view(Z, T, T), ops.pack_kn(v), T, d)
            o = o.view(B, H, T, d).permute(0, 2, 1, 3).reshape(B * T, C)
            x = ops.layernorm(ops.linear(o, L["ao"], res=x), *L["ln1"], eps=cfg["layer_norm_eps"])
            h = ops.linear(x, L["fi"], act=ops.ACT_GELU)
            x = ops.layernorm(ops.linear(h, L["fo"], res=x), *L["ln2"], eps=cfg["layer_norm_eps"])
        cls = x.view(B, T, C)[:, 0].contiguous()
        pooled = ops.linear(cls, pk["pool"], act=ops.ACT_TANH)               # RobertaPooler
        e = ops.linear(ops.linear(pooled, pk["p0"], act=ops.ACT_LRELU, act_slope=0.0), pk["p2"])   # :525-529 (ReLU)
        e = e.contiguous()
        return ops.rowscale_add(e, ops.row_l2norm(e, JOINT_DIM), divide=True)  # F.normalize(dim=-1)

    def get_audio_embedding(self, data):
        raise NotImplementedError("the HTSAT audio tower of CLAP is not built (SURVEY.md §8(f) rank 4)")


class CLAPAudioEmbeddingClassifierFreev2(nn.Module):
    def __init__(self, pretrained_path="", enable_cuda=False, sampling_rate=16000, embed_mode="audio", amodel="HTSAT-base",
                 unconditional_prob=0.1, random_mute=False, max_random_mute_portion=0.5, training_mode=True, config=None):
        super().__init__()
        self.device = "cuda"
        self.cuda = enable_cuda
        self.amodel, self.tmodel = amodel, "roberta"
        self.pretrained = pretrained_path
        self.embed_mode = self.embed_mode_orig = embed_mode
        self.sampling_rate = sampling_rate
        self.unconditional_prob = unconditional_prob
        self.random_mute, self.max_random_mute_portion, self.training_mode = random_mute, max_random_mute_portion, training_mode
        self.tokenize = None
        try:  # encoders/modules.py:573; offline the Hub is unreachable
            from transformers import RobertaTokenizer
            self.tokenize = RobertaTokenizer.from_pretrained("roberta-base")
        except Exception:
            pass
        self.model = CLAPTextModel(config)
        for p in self.model.parameters():
            p.requires_grad = False
        self.unconditional_token = None
        self.eval()

    def tokenizer(self, text):
        """encoders/modules.py:737-745"""
        if self.tokenize is None:
            raise RuntimeError("CLAP: RobertaTokenizer could not be loaded (no Hub access); tokenize elsewhere and call "
                               "encode_tokens(input_ids, attention_mask)")
        result = self.tokenize(text, padding="max_length", truncation=True, max_length=512, return_tensors="pt")
        return {k: v.squeeze(0) for k, v in result.items()}

    def build_unconditional_emb(self, tokens=None):
        """encoders/modules.py:655-658: the embedding of "" (first of a pair)."""
        data = tokens if tokens is not None else self.tokenizer(["", ""])
        self.unconditional_token = self.model.get_text_embedding(data)[0:1]

    def get_unconditional_condition(self, batchsize):
        """encoders/modules.py:606-610"""
        self.build_unconditional_emb()
        return torch.cat([self.unconditional_token.unsqueeze(0)] * batchsize, dim=0)

    def make_decision(self, probability):
        """encoders/modules.py:618-622: one host-generator uniform per call (RNG contract R2)."""
        return float(torch.rand(1)) < probability

    def encode_tokens(self, input_ids, attention_mask):
        """forward() of "text" mode after the tokenizer (encoders/modules.py:717-735): [B, 1, 512]."""
        if self.unconditional_token is None:
            raise RuntimeError("call build_unconditional_emb(tokens_of_empty_prompt) first (the reference does it lazily "
                               "through its tokenizer, encoders/modules.py:680-681)")
        embed = self.model.get_text_embedding({"input_ids": input_ids, "attention_mask": attention_mask}).unsqueeze(1)
        for i in range(embed.size(0)):
            if self.make_decision(self.unconditional_prob):
                embed[i] = self.unconditional_token
        return embed.detach()

    def forward(self, batch):
        if self.embed_mode == "audio":
            raise NotImplementedError("CLAP audio embedding needs the HTSAT tower (SURVEY.md §8(f) rank 4): not built")
        if self.unconditional_token is None:
            self.build_unconditional_emb()
        text_data = self.tokenizer(batch)
        if isinstance(batch, str) or (isinstance(batch, list) and len(batch) == 1):
            for key in text_data.keys():
                text_data[key] = text_data[key].unsqueeze(0)
        return self.encode_tokens(text_data["input_ids"], text_data["attention_mask"])

    def cos_similarity(self, waveform, text):
        raise NotImplementedError("CLAP re-ranking needs the HTSAT audio tower (SURVEY.md §8(f) rank 4): not built")
