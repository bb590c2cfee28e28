"""CLAP text tower (second half): the oracle against the fixture generated with transformers'
RobertaModel + the reference's projection head (CPU), state-dict compatibility (CPU), the HIP path against the fixture (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cases, clap_text, weights

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sd():
    with open(os.path.join(GOLD, "clap_text_keys.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    return weights.make_state_dict(shapes, seed=0), shapes


def _rel(a, b):
    a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_oracle_matches_the_transformers_fixture():
    g = np.load(os.path.join(GOLD, "clap_text_base2_b3.npz"))
    sd, _ = _sd()
    cfg = cases.clap_text_test_config()
    ids, mask = cases.clap_text_tokens()
    assert _rel(clap_text.roberta_pooled(sd, cfg, ids, mask), g["pooled"]) < 2e-6
    emb = clap_text.text_embedding(sd, cfg, ids, mask)
    assert _rel(emb, g["emb"]) < 2e-6
    assert torch.allclose(emb.norm(dim=-1), torch.ones(3), atol=1e-6)


def test_product_module_has_the_reference_state_dict():
    from audioldm2_amd.clap import CLAPAudioEmbeddingClassifierFreev2
    sd, shapes = _sd()
    m = CLAPAudioEmbeddingClassifierFreev2(embed_mode="text", config=cases.clap_text_test_config(), audio_config=False)
    assert {k: tuple(v.shape) for k, v in m.model.state_dict().items()} == shapes
    m.model.load_state_dict(sd, strict=True)
    with pytest.raises(RuntimeError):                    # built without its audio branch: says so instead of guessing
        m.model.get_audio_embedding({"waveform": torch.zeros(1, 48000)})
    # with the audio branch (the default, like the reference's create_model) the text keys are the same ones
    full = CLAPAudioEmbeddingClassifierFreev2(embed_mode="text", config=cases.clap_text_test_config(),
                                              audio_config=cases.htsat_test_config())
    have = {k: tuple(v.shape) for k, v in full.model.state_dict().items() if k.startswith("text_")}
    assert have == shapes


@pytest.mark.gpu
def test_hip_clap_text_tower_matches_fixture_and_replays_the_unconditional_draws():
    from audioldm2_amd.clap import CLAPAudioEmbeddingClassifierFreev2
    g = np.load(os.path.join(GOLD, "clap_text_base2_b3.npz"))
    sd, _ = _sd()
    m = CLAPAudioEmbeddingClassifierFreev2(embed_mode="text", unconditional_prob=0.0, config=cases.clap_text_test_config(),
                                           audio_config=False)
    m.model.load_state_dict(sd, strict=True)
    ids, mask = cases.clap_text_tokens()
    emb = m.model.get_text_embedding({"input_ids": ids, "attention_mask": mask})
    e = _rel(emb.cpu(), g["emb"])
    print(f"clap text tower vs fixture: {e:.2e}")
    assert e < 5e-5
    # forward() semantics after the tokenizer: [B, 1, 512]; rows replaced by the unconditional token with probability p,
    # one host-generator uniform per row (encoders/modules.py:731-733) — the same draws a CPU reference run consumes
    m.build_unconditional_emb({"input_ids": ids[2:3].repeat(2, 1), "attention_mask": mask[2:3].repeat(2, 1)})
    m.unconditional_prob = 0.5
    torch.manual_seed(5)
    out = m.encode_tokens(ids, mask)
    torch.manual_seed(5)
    draws = [float(torch.rand(1)) < 0.5 for _ in range(3)]
    assert tuple(out.shape) == (3, 1, 512)
    for i, replaced in enumerate(draws):
        ref = m.unconditional_token[0] if replaced else emb[i]
        assert torch.equal(out[i, 0], ref)
