"""Reference-side conditioner stack for the end-to-end conditioner fixture (TEST INFRA ONLY; build container only).

`oracle/make_golden.py e2econd` builds the REAL `ddpm.LatentDiffusion` whose `crossattn_audiomae_generated` conditioner is the
REAL `SequenceGenAudioMAECond` (encoders/modules.py:201-300, loaded from the reference file under a private module name: the
package-level module is pre-seeded with a stand-in by oracle/refimport.py) over
  * the REAL `FlanT5HiddenState` (encoders/modules.py:113-198) at flan-t5-large's geometry with 3 layers, its two Hub calls
    replaced (T5Config(**cfg), a stub tokenizer: cases.StubT5Tokenizer),
  * `RefClapText`: the CLAP text path restated around transformers' RobertaModel + the reference's projection head, exactly as
    tests/golden/clap_text_base2_b3 is produced (the reference's own class needs laion-clap's factory, Hub weights and the
    RoBERTa tokenizer) — forward / unconditional logic follows encoders/modules.py:606-610, 618-622, 655-735,
  * `RefAudioMAEStub`: the AudioMAE conditioner needs `timm`; on the sampling path only its unconditional condition (zeros,
    encoders/modules.py:465-479) is consumed, and its forward output is dropped (see audioldm2_amd/seqgen.py).
The targets below are what the fixture's cond_stage_config names; the reference's own instantiate_from_config builds them."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import cases, refimport


def encoders_modules():
    """The reference's encoders/modules.py executed under a private name (its heavy imports stubbed)."""
    refimport.install()
    import importlib.util
    import os
    import sys
    name = "_aldm_ref_encoders_modules"
    if name not in sys.modules:
        path = os.path.join(refimport.REF_ROOT, "audioldm2", "latent_diffusion", "modules", "encoders", "modules.py")
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    from transformers import GPT2Config
    GPT2Config.from_pretrained = classmethod(lambda cls, n, *a, **k: GPT2Config())   # sequence_input.py:69 (== "gpt2")
    return sys.modules[name]


def ref_flan_t5(**params):
    tok = cases.StubT5Tokenizer()
    m = refimport.flan_t5_hidden_state(cases.t5_test_config(), lambda prompt: (lambda r: (r.input_ids, r.attention_mask))(tok(prompt)))
    return m


class _ClapModel(nn.Module):
    def __init__(self):
        super().__init__()
        from transformers import RobertaConfig, RobertaModel
        self.text_branch = RobertaModel(RobertaConfig(**cases.clap_text_test_config())).eval()
        self.text_projection = nn.Sequential(nn.Linear(768, 512), nn.ReLU(), nn.Linear(512, 512)).eval()

    def get_text_embedding(self, data):
        """clap/open_clip/model.py:656-663, 730-747"""
        x = self.text_branch(input_ids=data["input_ids"], attention_mask=data["attention_mask"])["pooler_output"]
        return F.normalize(self.text_projection(x), dim=-1)


class RefClapText(nn.Module):
    def __init__(self, sampling_rate=48000, embed_mode="text", amodel="HTSAT-base", unconditional_prob=0.1, **kw):
        super().__init__()
        assert embed_mode == "text"
        self.model = _ClapModel()
        self.unconditional_prob = unconditional_prob
        self.unconditional_token = None
        self.tokenize = cases.StubRobertaTokenizer()

    def tokenizer(self, text):   # encoders/modules.py:737-745
        result = self.tokenize(text, padding="max_length", truncation=True, max_length=512, return_tensors="pt")
        return {k: v.squeeze(0) for k, v in result.items()}

    def build_unconditional_emb(self):   # :655-658
        self.unconditional_token = self.model.get_text_embedding(self.tokenizer(["", ""]))[0:1]

    def get_unconditional_condition(self, batchsize):   # :606-610
        self.build_unconditional_emb()
        return torch.cat([self.unconditional_token.unsqueeze(0)] * batchsize, dim=0)

    @torch.no_grad()
    def forward(self, batch):   # :660-735, text mode
        if self.unconditional_token is None:
            self.build_unconditional_emb()
        text_data = self.tokenizer(batch)
        if isinstance(batch, str) or (isinstance(batch, list) and len(batch) == 1):
            for key in text_data.keys():
                text_data[key] = text_data[key].unsqueeze(0)
        embed = self.model.get_text_embedding(text_data).unsqueeze(1)
        for i in range(embed.size(0)):
            if float(torch.rand(1)) < self.unconditional_prob:   # make_decision, :618-622
                embed[i] = self.unconditional_token
        return embed.detach()


class RefAudioMAEStub(nn.Module):
    def __init__(self, eval_time_pooling=None, eval_freq_pooling=None, **kw):
        super().__init__()
        self.n = int(512 / (min(eval_time_pooling, 64) * min(eval_freq_pooling, 8)))

    def get_unconditional_condition(self, batchsize):   # encoders/modules.py:465-479
        return [torch.zeros((batchsize, self.n, 768)), torch.ones((batchsize, self.n))]

    def forward(self, batch):
        return self.get_unconditional_condition(batch.shape[0])


def cond_stage_config():
    """audioldm2-full's cond_stage_config (utils.py:354-411) with the targets above."""
    inner = {
        "film_clap_cond1": {"cond_stage_key": "text", "conditioning_key": "film", "target": "oracle.refcond.RefClapText",
                            "params": {"sampling_rate": 48000, "embed_mode": "text", "amodel": "HTSAT-base"}},
        "crossattn_flan_t5": {"cond_stage_key": "text", "conditioning_key": "crossattn", "target": "oracle.refcond.ref_flan_t5"},
        "crossattn_audiomae_pooled": {"cond_stage_key": "ta_kaldi_fbank", "conditioning_key": "crossattn",
                                      "target": "oracle.refcond.RefAudioMAEStub",
                                      "params": {"regularization": False, "no_audiomae_mask": True, "time_pooling_factors": [8],
                                                 "freq_pooling_factors": [8], "eval_time_pooling": 8, "eval_freq_pooling": 8,
                                                 "mask_ratio": 0}}}
    return {
        "crossattn_audiomae_generated": {
            "cond_stage_key": "all", "conditioning_key": "crossattn",
            "target": "_aldm_ref_encoders_modules.SequenceGenAudioMAECond",
            "params": {"always_output_audiomae_gt": False, "learnable": True, "device": "cpu", "use_gt_mae_output": True,
                       "use_gt_mae_prob": 0.0, "base_learning_rate": 0.0002, "sequence_gen_length": 8, "use_warmup": True,
                       "sequence_input_key": ["film_clap_cond1", "crossattn_flan_t5"], "sequence_input_embed_dim": [512, 1024],
                       "batchsize": 16, "cond_stage_config": inner}},
        "crossattn_flan_t5": {"cond_stage_key": "text", "conditioning_key": "crossattn", "target": "oracle.refcond.ref_flan_t5"}}
