"""AudioMAE-token sequence generator on the MI355X.  Host logic verified on the CPU against the
reference fixtures with the ops replaced by torch stand-ins (tests/test_host_logic.py); the HIP path against the same fixtures
in tests/test_seqgen_gpu.py.

Mirrors `Sequence2AudioMAE` (audioldm2/audiomae_gen/sequence_input.py): same constructor keywords, same state-dict keys
(`start_of_sequence_tokens`, `end_of_sequence_tokens`, `input_sequence_embed_linear.{i}`, `model.{wte,wpe,h.{l}.*,ln_f}` with
GPT-2's `Conv1D` weights stored [in, out]) and the same `generate(batch, cond_dict) -> (tokens [B, steps, 768], cond_dict)`.

The reference re-runs GPT-2 over the whole prefix for every generated token (sequence_input.py:308-323; 512 full forwards
for the speech model).  Here the prefix is run once and every later position attends to cached keys / values — the same
function (oracle/seqgen.py: cached == full re-forward), O(n) instead of O(n^2) forwards.  The cache has a fixed length
(prefix + steps, rounded up to 4): positions that do not exist yet are masked, so every step launches identical shapes —
and from the third generated token on the step is ONE HIP graph replay: the position lives in a device tensor (embedding
row, cache slot and key-mask entry are addressed through it, and a captured add advances it), so nothing in the launch
sequence changes from token to token.  A decode step is ~160 launches of microsecond kernels (M = batch rows): launched from
Python it costs 2.4 ms per token — 1.2 s for the speech model's 512 tokens, a third of that config's sampling job — replayed
as a graph it is bound by the kernels (tools/cond_bench.py, profiles/r02_cond_bench.txt): 1.8 ms per token at batch 8.
Round 4: at batch <= 16 the decode step's GPT-2 blocks run on the single-position kernels of csrc/decode.hip
(`aldm_decode_linear`: LayerNorm + Conv1D weight stream + bias / tanh-GELU / residual in one launch, exact fp32 FMA;
`aldm_decode_attention`: cache append + scores + softmax + P.V per (sample, head)): 5 launches per block instead of ~17
(`ALDM_SEQGEN_DECODE=general` selects the old path for A/B).
All arithmetic goes through the C ABI: `aldm_layernorm`, `aldm_igemm` (projections with fused bias / tanh-GELU /
residual; Q·K^T as the NT batched product, P·V through `aldm_pack_kn`), `aldm_softmax_rows_masked`, `aldm_axpby`;
torch only moves data (concatenation, head split / merge copies, cache writes).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_GELU_TANH

N_LAYER, N_HEAD, N_EMBD, HEAD_DIM, N_POS, LN_EPS = 12, 12, 768, 64, 1024, 1e-5  # GPT2Config defaults == "gpt2"


class _Conv1D(nn.Module):
    """transformers' Conv1D parameter layout: y = x @ weight + bias with weight [in, out]."""

    def __init__(self, n_in: int, n_out: int):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(n_in, n_out) * 0.02)
        self.bias = nn.Parameter(torch.zeros(n_out))


class _LN(nn.Module):
    def __init__(self, n: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))
        self.bias = nn.Parameter(torch.zeros(n))


class _Attn(nn.Module):
    def __init__(self):
        super().__init__()
        self.c_attn = _Conv1D(N_EMBD, 3 * N_EMBD)
        self.c_proj = _Conv1D(N_EMBD, N_EMBD)


class _MLP(nn.Module):
    def __init__(self):
        super().__init__()
        self.c_fc = _Conv1D(N_EMBD, 4 * N_EMBD)
        self.c_proj = _Conv1D(4 * N_EMBD, N_EMBD)


class _Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.ln_1, self.attn, self.ln_2, self.mlp = _LN(N_EMBD), _Attn(), _LN(N_EMBD), _MLP()


class _GPT2(nn.Module):
    """Parameter container with GPT2Model's names; `wte` is never used (the generator feeds inputs_embeds) but is part of
    the reference's state dict."""

    def __init__(self):
        super().__init__()
        self.wte = nn.Embedding(50257, N_EMBD)
        self.wpe = nn.Embedding(N_POS, N_EMBD)
        self.h = nn.ModuleList([_Block() for _ in range(N_LAYER)])
        self.ln_f = _LN(N_EMBD)


def _f(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(dtype=torch.float32).contiguous()


class Sequence2AudioMAE(nn.Module):
    def __init__(self, base_learning_rate=None, sequence_gen_length: int = 8, sequence_input_key: Sequence[str] = (),
                 sequence_input_embed_dim: Sequence[int] = (), cond_stage_config: Optional[dict] = None, **kwargs):
        super().__init__()
        self.mae_token_num = int(sequence_gen_length)
        self.sequence_input_key = list(sequence_input_key)
        self.sequence_input_embed_dim = list(sequence_input_embed_dim)
        self.start_of_sequence_tokens = nn.Embedding(32, N_EMBD)
        self.end_of_sequence_tokens = nn.Embedding(32, N_EMBD)
        self.input_sequence_embed_linear = nn.ModuleList([nn.Linear(d, N_EMBD) for d in self.sequence_input_embed_dim])
        self.model = _GPT2()
        # conditioner sub-modules through the reference's plugin seam (sequence_input.py:372-383)
        self.cond_stage_models = nn.ModuleList([])
        self.cond_stage_model_metadata = {}
        if cond_stage_config:
            from .pipeline import instantiate_from_config
            for i, (key, cfg) in enumerate(cond_stage_config.items()):
                self.cond_stage_models.append(instantiate_from_config(cfg))
                self.cond_stage_model_metadata[key] = {"model_idx": i, "cond_stage_key": cfg["cond_stage_key"],
                                                       "conditioning_key": cfg["conditioning_key"]}
        self._pk = None
        self.eval()

    # ---- packed weights (built once; invalidate_packed() after loading new weights) ---------------------------------
    def invalidate_packed(self):
        self._pk = None

    def load_state_dict(self, *args, **kwargs):
        """New weights invalidate every packed copy (prefill images and decode operands alike)."""
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_packed()
        return out

    def _packed(self):
        if self._pk is None:
            def conv1d(m):  # Conv1D [in, out] -> the Linear layout pack_conv expects
                return ops.pack_conv(m.weight.detach().t().contiguous(), m.bias)
            def kn(m):      # Conv1D's own [in, out] weight + bias: the operand of the single-position decode kernels — a COPY, like
                # every other packed operand: aliasing the live parameter would let an in-place weight update reach the decode path
                # but not the (packed) prefill path
                return (_f(m.weight).clone(), _f(m.bias).clone())
            blocks = []
            for b in self.model.h:
                blocks.append(dict(ln1=(_f(b.ln_1.weight), _f(b.ln_1.bias)), ln2=(_f(b.ln_2.weight), _f(b.ln_2.bias)),
                                   c_attn=conv1d(b.attn.c_attn), c_proj=conv1d(b.attn.c_proj),
                                   c_fc=conv1d(b.mlp.c_fc), m_proj=conv1d(b.mlp.c_proj),
                                   kn_attn=kn(b.attn.c_attn), kn_proj=kn(b.attn.c_proj), kn_fc=kn(b.mlp.c_fc),
                                   kn_mproj=kn(b.mlp.c_proj)))
            self._pk = dict(blocks=blocks, ln_f=(_f(self.model.ln_f.weight), _f(self.model.ln_f.bias)),
                            inp=[ops.pack_conv(l.weight, l.bias) for l in self.input_sequence_embed_linear],
                            wpe=_f(self.model.wpe.weight), sos=_f(self.start_of_sequence_tokens.weight),
                            eos=_f(self.end_of_sequence_tokens.weight))
        return self._pk

    # ---- sequence_input.py:136-199 (+ :109-124) -------------------------------------------------------------------
    def get_input_sequence_and_mask(self, cond_dict: Dict[str, object]) -> Tuple[torch.Tensor, torch.Tensor, int]:
        pk = self._packed()
        embeds, masks = [], []
        for i, key in enumerate(self.sequence_input_key):
            assert key in cond_dict, "Invalid sequence key %s" % key
            c = cond_dict[key]
            if isinstance(c, (list, tuple)):
                assert len(c) == 2, "The crossattn returned list should have length 2, including embed and attn_mask"
                x, m = c
            else:
                x, m = c, torch.ones((c.shape[0], c.shape[1]), device=c.device)
            x = ops.linear(x.to(torch.float32).contiguous(), pk["inp"][i])
            B = x.shape[0]
            one = torch.ones((B, 1), device=x.device)
            embeds += [pk["sos"][i].expand(B, 1, -1), x, pk["eos"][i].expand(B, 1, -1)]
            masks += [one, m.to(torch.float32), one]
        x, m = torch.cat(embeds, dim=1), torch.cat(masks, dim=1)
        max_len = N_POS - self.mae_token_num
        if x.shape[1] > max_len:
            print("The input sequence length to GPT-2 model is too long:", x.shape[1])
            x, m = x[:, :max_len], m[:, :max_len]
        return x.contiguous(), m.contiguous(), x.shape[1]

    # ---- GPT-2 over T new positions starting at pos0, keys / values of every position in fixed-length caches -------
    def _forward_positions(self, x: torch.Tensor, pos0: int, kc: List[torch.Tensor], vc: List[torch.Tensor],
                           keymask: torch.Tensor) -> torch.Tensor:
        pk = self._packed()
        B, T, _ = x.shape
        Z, n_tot = B * N_HEAD, keymask.shape[1]
        pos = pk["wpe"][pos0:pos0 + T].expand(B, T, N_EMBD).contiguous()
        h = ops.axpby(x.contiguous(), pos, 1.0, 1.0).view(B * T, N_EMBD)
        for l, blk in enumerate(pk["blocks"]):
            a = ops.layernorm(h, blk["ln1"][0], blk["ln1"][1], LN_EPS)
            qkv = ops.linear(a, blk["c_attn"]).view(B, T, 3, N_HEAD, HEAD_DIM)
            q = qkv[:, :, 0].permute(0, 2, 1, 3).reshape(Z, T, HEAD_DIM)              # head split: copies
            kc[l].view(B, N_HEAD, n_tot, HEAD_DIM)[:, :, pos0:pos0 + T] = qkv[:, :, 1].permute(0, 2, 1, 3)
            vc[l].view(B, N_HEAD, n_tot, HEAD_DIM)[:, :, pos0:pos0 + T] = qkv[:, :, 2].permute(0, 2, 1, 3)
            s = ops.gemm_nt(q, kc[l], alpha=1.0 / math.sqrt(HEAD_DIM))               # [Z, T, n_tot]
            p = ops.softmax_rows_masked(s.view(B, N_HEAD, T, n_tot), keymask, pos0)
            o = ops.gemm_packed_batched(p.view(Z, T, n_tot), ops.pack_kn(vc[l]), n_tot, HEAD_DIM)  # [Z, T, 64]
            o = o.view(B, N_HEAD, T, HEAD_DIM).permute(0, 2, 1, 3).reshape(B * T, N_EMBD)           # head merge: copy
            h = ops.linear(o, blk["c_proj"], res=h)
            m = ops.layernorm(h, blk["ln2"][0], blk["ln2"][1], LN_EPS)
            m = ops.linear(m, blk["c_fc"], act=ACT_GELU_TANH)
            h = ops.linear(m, blk["m_proj"], res=h)
        return ops.layernorm(h, pk["ln_f"][0], pk["ln_f"][1], LN_EPS).view(B, T, N_EMBD)

    # ---- one new position whose index lives on the device (the graph-replayed decode step) ---------------------------
    def _decode_one(self, tok: torch.Tensor, pos: torch.Tensor, kc: List[torch.Tensor], vc: List[torch.Tensor],
                    keymask: torch.Tensor) -> torch.Tensor:
        """Same arithmetic as `_forward_positions(tok, pos, ...)` for T = 1, with every use of the position going through
        the one-element int64 device tensor `pos`: embedding row by index_select, cache slot by index_copy_, and no causal
        limit in the softmax (the key mask already excludes every later position: they have not been switched on yet)."""
        pk = self._packed()
        B = tok.shape[0]
        Z, n_tot = B * N_HEAD, keymask.shape[1]
        wpe = pk["wpe"].index_select(0, pos).expand(B, 1, N_EMBD).contiguous()
        h = ops.axpby(tok.contiguous(), wpe, 1.0, 1.0).view(B, N_EMBD)
        mode = getattr(self, "_decode_fast", None)
        if mode is None:   # (direct callers; generate() reads the switch once per call)
            mode = os.environ.get("ALDM_SEQGEN_DECODE", "split")
        if B <= ops.DECODE_MAX_ROWS and mode == "split":
            # round 6: every launch on the whole chip — K cut into slices (one HBM round trip per block), the partial sums handed to
            # the next launch (csrc/decode.hip, decode_gemv / decode_reduce_ln): 7 launches per block, each a fraction of the
            # column-tile kernels' time (those give 24 - 96 blocks all of K: one compute unit's memory path per 96 - 393 KB)
            blocks = pk["blocks"]
            h, xn = ops.decode_reduce_ln(None, bias=pk["wpe"], bias_row=pos, res=tok.contiguous().view(B, N_EMBD),
                                         ln=(*blocks[0]["ln1"], LN_EPS))
            for l, blk in enumerate(blocks):
                qp = ops.decode_gemv(xn, blk["kn_attn"][0])
                o = ops.decode_attention_parts(qp, blk["kn_attn"][1], pos, kc[l], vc[l], keymask, N_HEAD)
                pp = ops.decode_gemv(o, blk["kn_proj"][0])
                h, xn = ops.decode_reduce_ln(pp, bias=blk["kn_proj"][1], res=h, ln=(*blk["ln2"], LN_EPS))
                fp = ops.decode_gemv(xn, blk["kn_fc"][0])
                mp = ops.decode_gemv(fp, blk["kn_mproj"][0], xbias=blk["kn_fc"][1], xact=ACT_GELU_TANH)
                if l + 1 < len(blocks):
                    h, xn = ops.decode_reduce_ln(mp, bias=blk["kn_mproj"][1], res=h, ln=(*blocks[l + 1]["ln1"], LN_EPS))
                else:
                    xn = ops.decode_reduce_ln(mp, bias=blk["kn_mproj"][1], res=h, ln=(*pk["ln_f"], LN_EPS), want_h=False)
            return xn.view(B, 1, N_EMBD)
        if B <= ops.DECODE_MAX_ROWS and mode != "general":
            # B rows per Linear: weight streams, 5 launches per block (csrc/decode.hip) instead of the general path's ~17
            for l, blk in enumerate(pk["blocks"]):
                qkv = ops.decode_linear(h, *blk["kn_attn"], ln=(*blk["ln1"], LN_EPS))
                o = ops.decode_attention(qkv, pos, kc[l], vc[l], keymask, N_HEAD)
                h = ops.decode_linear(o, *blk["kn_proj"], res=h)
                m = ops.decode_linear(h, *blk["kn_fc"], ln=(*blk["ln2"], LN_EPS), act=ACT_GELU_TANH)
                h = ops.decode_linear(m, *blk["kn_mproj"], res=h)
            return ops.layernorm(h, pk["ln_f"][0], pk["ln_f"][1], LN_EPS).view(B, 1, N_EMBD)
        for l, blk in enumerate(pk["blocks"]):
            a = ops.layernorm(h, blk["ln1"][0], blk["ln1"][1], LN_EPS)
            qkv = ops.linear(a, blk["c_attn"]).view(B, 1, 3, N_HEAD, HEAD_DIM)
            q = qkv[:, :, 0].permute(0, 2, 1, 3).reshape(Z, 1, HEAD_DIM)
            kc[l].view(B, N_HEAD, n_tot, HEAD_DIM).index_copy_(2, pos, qkv[:, :, 1].permute(0, 2, 1, 3))
            vc[l].view(B, N_HEAD, n_tot, HEAD_DIM).index_copy_(2, pos, qkv[:, :, 2].permute(0, 2, 1, 3))
            s = ops.gemm_nt(q, kc[l], alpha=1.0 / math.sqrt(HEAD_DIM))
            p = ops.softmax_rows_masked(s.view(B, N_HEAD, 1, n_tot), keymask, n_tot)
            o = ops.gemm_packed_batched(p.view(Z, 1, n_tot), ops.pack_kn(vc[l]), n_tot, HEAD_DIM)
            o = o.view(B, N_HEAD, 1, HEAD_DIM).permute(0, 2, 1, 3).reshape(B, N_EMBD)
            h = ops.linear(o, blk["c_proj"], res=h)
            m = ops.layernorm(h, blk["ln2"][0], blk["ln2"][1], LN_EPS)
            m = ops.linear(m, blk["c_fc"], act=ACT_GELU_TANH)
            h = ops.linear(m, blk["m_proj"], res=h)
        return ops.layernorm(h, pk["ln_f"][0], pk["ln_f"][1], LN_EPS).view(B, 1, N_EMBD)

    # ---- sequence_input.py:294-325 -------------------------------------------------------------------------------
    GRAPH_MIN_STEPS = 16  # shorter generations (the 8-token text-to-audio configs) are not worth a capture

    @torch.no_grad()
    def generate(self, batch, cond_dict: Optional[dict] = None, no_grad: bool = False):
        if cond_dict is None:
            cond_dict = self.get_input(batch)
        x, mask, P = self.get_input_sequence_and_mask(cond_dict)
        self._decode_fast = os.environ.get("ALDM_SEQGEN_DECODE", "split")   # "split" | "fast" | "general"; read once per generation
        B, steps = x.shape[0], self.mae_token_num
        n_tot = (P + steps + 3) // 4 * 4
        dev = x.device
        kc = [torch.zeros((B * N_HEAD, n_tot, HEAD_DIM), device=dev) for _ in range(N_LAYER)]
        vc = [torch.zeros((B * N_HEAD, n_tot, HEAD_DIM), device=dev) for _ in range(N_LAYER)]
        keymask = torch.zeros((B, n_tot), device=dev)
        keymask[:, :P] = mask
        out = self._forward_positions(x, 0, kc, vc, keymask)
        tok = out[:, -1:, :].contiguous()
        # every later token: one decode step that reads / advances device-side state only.  Long generations (the speech
        # model: 512 tokens) replay it as one HIP graph (eager the first time, captured the second: ddim.GraphStepper);
        # the 8-token text-to-audio configurations run the same step eagerly — a capture is not worth 7 steps
        from .ddim import GraphStepper
        toks = torch.empty((B, steps, N_EMBD), device=dev)
        toks[:, 0:1] = tok
        ps = torch.tensor([P, 1], device=dev, dtype=torch.long)      # (position, output slot): advanced together, one launch per token
        st = {"tok": tok.clone(), "pos": ps[0:1], "slot": ps[1:2]}

        def step(e=st):
            keymask.index_fill_(1, e["pos"], 1.0)        # the token joins the sequence (its own key is visible to it)
            new = self._decode_one(e["tok"], e["pos"], kc, vc, keymask)
            e["tok"].copy_(new)
            toks.index_copy_(1, e["slot"], new)
            ps.add_(1)
        run = GraphStepper(step, use_graph=x.is_cuda and steps >= self.GRAPH_MIN_STEPS and
                           os.environ.get("ALDM_NO_GRAPH", "0") != "1")
        try:
            for _ in range(steps - 1):
                run()
        finally:
            run.fn = None          # break the closure cycle: the graph and its pool go with this call, not with a later collection
            run.graph = None
            self._decode_fast = None   # a later direct `_decode_one` reads $ALDM_SEQGEN_DECODE again
        return toks, cond_dict

    # ---- conditioning through the sub-modules (sequence_input.py:327-370, 385-403) ---------------------------------
    @staticmethod
    def get_input_item(batch, k):
        """sequence_input.py:327-350: the view of the batch a conditioner with `cond_stage_key` k receives."""
        ret = {"fbank": batch["log_mel_spec"].unsqueeze(1).contiguous().float(), "stft": batch["stft"].contiguous().float(),
               "waveform": batch["waveform"].contiguous().float(), "text": list(batch["text"]), "fname": batch["fname"]}
        for key in batch.keys():
            if key not in ret:
                ret[key] = batch[key]
        return ret[k]

    def get_input(self, batch) -> dict:
        """sequence_input.py:352-370 at unconditional_cfg = False: every sub-conditioner on its view of the batch."""
        cond = {}
        for key, meta in self.cond_stage_model_metadata.items():
            xc = self.get_input_item(batch, meta["cond_stage_key"])
            if torch.is_tensor(xc) and torch.cuda.is_available():
                xc = xc.cuda()
            cond[key] = self.cond_stage_models[meta["model_idx"]](xc)
        return cond

    def cfg_uncond(self, batch_size):
        """sequence_input.py:85-98: the unconditional condition of every sub-conditioner; the (never generated) CLAP-to-AudioMAE
        feature is the pooled AudioMAE one."""
        uncond = {}
        for key, meta in self.cond_stage_model_metadata.items():
            uncond[key] = self.cond_stage_models[meta["model_idx"]].get_unconditional_condition(batch_size)
        assert "crossattn_audiomae_pooled" in uncond, "The module is not initialized with AudioMAE"
        uncond["crossattn_clap_to_audiomae_feature"] = uncond["crossattn_audiomae_pooled"]
        return uncond


class SequenceGenAudioMAECond(Sequence2AudioMAE):
    """Drop-in for `audioldm2.latent_diffusion.modules.encoders.modules.SequenceGenAudioMAECond` (encoders/modules.py:201-300),
    the `cond_stage_config` target of `crossattn_audiomae_generated` (utils.py:127, :354): same constructor keywords, same
    state-dict keys (Sequence2AudioMAE's + `cond_stage_models.{i}.*` of the sub-conditioners), `forward(batch) -> dict` and
    `get_unconditional_condition(batchsize) -> dict`, so the reference's LatentDiffusion wires it through the config string
    alone.  At inference the reference always generates (the ground-truth AudioMAE branch is commented
    out, encoders/modules.py:275-278)."""

    def __init__(self, cond_stage_config, base_learning_rate, sequence_gen_length, sequence_input_key, sequence_input_embed_dim,
                 batchsize, always_output_audiomae_gt=False, pretrained_path=None, force_reload_pretrain_avoid_overwrite=False,
                 learnable=True, use_warmup=True, device=None, use_gt_mae_output=None, use_gt_mae_prob=None):
        super().__init__(base_learning_rate=base_learning_rate, cond_stage_config=cond_stage_config,
                         sequence_gen_length=sequence_gen_length, sequence_input_key=sequence_input_key,
                         sequence_input_embed_dim=sequence_input_embed_dim, use_warmup=False, batchsize=batchsize)
        assert use_gt_mae_output is not None and use_gt_mae_prob is not None
        self.always_output_audiomae_gt = always_output_audiomae_gt
        self.force_reload_pretrain_avoid_overwrite = force_reload_pretrain_avoid_overwrite
        self.pretrained_path = pretrained_path
        self.device = device
        self.is_reload = not force_reload_pretrain_avoid_overwrite
        self.load_pretrain_model()
        self.use_gt_mae_output, self.use_gt_mae_prob, self.learnable = use_gt_mae_output, use_gt_mae_prob, learnable
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())
        for p in self.parameters():   # sampling only: nothing here trains
            p.requires_grad = False
        self.eval()

    def load_pretrain_model(self):
        """encoders/modules.py:258-262"""
        if self.pretrained_path is not None:
            print("Reload SequenceGenAudioMAECond from %s" % self.pretrained_path)
            self.load_state_dict(torch.load(self.pretrained_path)["state_dict"])
            self.invalidate_packed()

    def get_unconditional_condition(self, batchsize):
        """encoders/modules.py:265-271: a DICT — DiffusionWrapper.route takes its last crossattn entry, the zero tokens."""
        ret = self.cfg_uncond(batchsize)
        pooled = ret["crossattn_audiomae_pooled"]
        ret["crossattn_audiomae_generated"] = [pooled[0], torch.ones_like(pooled[1]).float()]
        return ret

    @torch.no_grad()
    def forward(self, batch):
        """encoders/modules.py:273-300: generate the AudioMAE tokens from the batch's text conditions; the sub-conditioners'
        outputs ride along under their own keys (LatentDiffusion keeps the ones that are its cond keys, e.g. nothing else
        for audioldm2-full — its own `crossattn_flan_t5` model runs again, like the reference's)."""
        if self.force_reload_pretrain_avoid_overwrite and not self.is_reload:
            self.load_pretrain_model()
            self.is_reload = True
        tokens, cond_dict = self.generate(batch)
        mask = torch.ones((tokens.size(0), tokens.size(1)), device=tokens.device).float()
        ret = {"crossattn_audiomae_generated": [tokens, mask]}
        for key in cond_dict.keys():
            ret[key] = cond_dict[key]
        return ret


class AudioMAEConditionCTPoolRand(nn.Module):
    """Sampling-path stand-in for `encoders/modules.py:427-546` (AudioMAE ViT + pooling).  On the sampling path this conditioner
    contributes exactly two things, both reproduced here: (1) `get_unconditional_condition` — zero tokens + a ones mask of
    512 / (eval_time_pooling * eval_freq_pooling) positions (encoders/modules.py:465-479), which SequenceGenAudioMAECond
    turns into the unconditional `crossattn_audiomae_generated`; (2) a `crossattn_audiomae_pooled` entry in the dict
    `SequenceGenAudioMAECond.forward` returns — computed by the reference from `ta_kaldi_fbank` (all zeros in
    `make_batch_for_text_to_audio`, pipeline.py:116) and consumed by NOBODY at inference: it is neither a sequence input key
    of the generator nor a cond key of LatentDiffusion (utils.py:354-411).  forward() therefore returns tokens of the right
    shape without running a ViT (the reference needs `timm` for it; out of the hot path, SURVEY §2 OUT-OF-SCOPE).  The
    checkpoint's `audiomae.*` tensors have no home here and are reported as unused by load_reference_state_dict."""

    def __init__(self, time_pooling_factors=(1, 2, 4, 8), freq_pooling_factors=(1, 2, 4, 8), eval_time_pooling=None,
                 eval_freq_pooling=None, mask_ratio=0.0, regularization=False, no_audiomae_mask=True,
                 no_audiomae_average=False):
        super().__init__()
        self.eval_time_pooling, self.eval_freq_pooling = eval_time_pooling, eval_freq_pooling
        self.time_pooling_factors, self.freq_pooling_factors = list(time_pooling_factors), list(freq_pooling_factors)
        self.mask_ratio, self.use_reg = mask_ratio, regularization
        self.no_audiomae_mask, self.no_audiomae_average = no_audiomae_mask, no_audiomae_average

    def _tokens(self, batchsize):
        token_num = int(512 / (min(self.eval_time_pooling, 64) * min(self.eval_freq_pooling, 8)))
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        return [torch.zeros((batchsize, token_num, 768), device=dev), torch.ones((batchsize, token_num), device=dev)]

    def get_unconditional_condition(self, batchsize):
        return self._tokens(batchsize)

    def forward(self, batch):
        return self._tokens(batch.shape[0])
