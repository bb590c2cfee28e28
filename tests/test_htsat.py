"""CLAP audio tower / candidate re-ranking: the oracle against the fixture generated with the REAL
`HTSAT_Swin_Transformer` (CPU), the product's host-side index logic and state dict (CPU), every glue kernel and the HIP tower
against the oracle / the fixture (GPU)."""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cases
from oracle import htsat as oh

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sd():
    with open(os.path.join(GOLD, "htsat_keys.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    return cases.htsat_state_dict(shapes), shapes


def _rel(a, b):
    a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _fixture():
    return np.load(os.path.join(GOLD, "htsat_base2222_b2.npz"))


# ---- CPU ----------------------------------------------------------------------------------------------------------------
def test_oracle_matches_the_real_htsat_fixture():
    g = _fixture()
    sd, _ = _sd()
    wav16 = cases.clap_waveform(2)
    wav48 = oh.resample(wav16, 16000, 48000)[:, :480000]
    assert wav48.shape == (2, 480000)
    assert _rel(wav48[:, :4096], g["wav48_head"]) < 1e-6
    e = oh.htsat_embedding(sd, wav48, cases.htsat_test_config())
    assert _rel(e, g["embedding"]) < 5e-6
    emb = oh.audio_embedding(sd, wav16, 16000, cases.htsat_test_config())
    assert _rel(emb, g["emb"]) < 5e-6
    assert torch.allclose(emb.norm(dim=-1), torch.ones(2), atol=1e-6)


def test_resampler_restates_a_polyphase_sinc_filter():
    """Properties of torchaudio's resampler the restatement must keep: DC gain ~ 1, a 1 kHz tone comes out as a 1 kHz tone
    at the new rate, length = ceil(new * T / orig)."""
    x = torch.ones(1, 4000)
    y = oh.resample(x, 16000, 48000)
    assert y.shape == (1, 12000)
    assert float((y[0, 300:-300] - 1).abs().max()) < 2e-3
    t = torch.arange(16000) / 16000.0
    y = oh.resample(torch.sin(2 * math.pi * 1000 * t)[None], 16000, 48000)[0]
    t48 = torch.arange(48000) / 48000.0
    assert float((y[600:-600] - torch.sin(2 * math.pi * 1000 * t48)[600:-600]).abs().max()) < 2e-3
    assert oh.resample(torch.zeros(1, 1001), 16000, 22050).shape[1] == math.ceil(22050 * 1001 / 16000)


def test_host_index_logic_of_the_product_matches_the_oracle():
    from audioldm2_amd import clap as pc
    for ws in (4, 8):
        assert torch.equal(pc._relative_position_index(ws), oh.relative_position_index(ws))
    for H, ws, shift in ((16, 8, 4), (16, 8, 0), (64, 8, 4), (8, 8, 0), (32, 8, 4)):
        x = torch.arange(H * H, dtype=torch.float32).view(1, H, H, 1)
        h = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2)) if shift else x
        want = oh.window_partition(h, ws).reshape(-1).long()
        idx = pc._window_index(H, H, ws, shift)
        assert torch.equal(idx, want)
        # the inverse permutation = window_reverse + roll back
        win = x.view(-1)[idx].view(-1, ws, ws, 1)
        back = oh.window_reverse(win, ws, H, H)
        back = torch.roll(back, shifts=(shift, shift), dims=(1, 2)) if shift else back
        inv = torch.empty_like(idx)
        inv[idx] = torch.arange(idx.numel())
        assert torch.equal(x.view(-1)[idx][inv], back.reshape(-1))
        if shift:
            assert torch.equal(pc._shift_attn_mask(H, H, ws, shift), oh.shift_attn_mask(H, H, ws, shift))
    k, width, down, up = pc.sinc_resample_kernel(16000, 48000)
    ko, wo, do, uo = oh.sinc_resample_kernel(16000, 48000)
    assert (width, down, up) == (wo, do, uo) == (7, 1, 3)
    assert torch.equal(k, ko[:, 0])


def test_product_module_holds_the_reference_state_dict():
    from audioldm2_amd.clap import AUDIO_CFG, HTSAT_BASE, CLAPAudioEmbeddingClassifierFreev2
    assert HTSAT_BASE == oh.HTSAT_BASE
    assert {k: AUDIO_CFG[k] for k in oh.AUDIO_CFG} == oh.AUDIO_CFG
    sd, shapes = _sd()
    m = CLAPAudioEmbeddingClassifierFreev2(embed_mode="audio", config=cases.clap_text_test_config(),
                                           audio_config=cases.htsat_test_config())
    have = {k: tuple(v.shape) for k, v in m.model.state_dict().items()}
    for k, shp in shapes.items():
        assert have[k] == shp, k
    # what the real class registers besides (oracle/make_golden.py: gen_htsat's `skip`) is there too
    for k in ("audio_branch.bn0.num_batches_tracked", "audio_branch.layers.0.blocks.0.attn.relative_position_index",
              "audio_branch.layers.0.blocks.1.attn_mask", "audio_branch.tscam_conv.weight", "audio_branch.head.weight"):
        assert k in have, k
    assert have["audio_branch.layers.0.blocks.1.attn_mask"] == (64, 64, 64)
    assert have["audio_branch.tscam_conv.weight"] == (527, 1024, 2, 3)
    assert "audio_branch.layers.3.blocks.1.attn_mask" not in have      # 8x8 resolution = one window: never shifted
    missing = m.model.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if k in shapes] and not missing.unexpected_keys
    # the restated torchlibrosa tensors are the oracle's
    st = m.model.audio_branch.spectrogram_extractor.stft
    from oracle import stft as ostft
    basis = torch.from_numpy(ostft.stft_forward_basis(1024, 1024))
    assert torch.equal(torch.cat([st.conv_real.weight[:, 0], st.conv_imag.weight[:, 0]]), basis)
    melW = torch.from_numpy(ostft.mel_filterbank(48000, 1024, 64, 50, 14000)).t()
    assert torch.equal(m.model.audio_branch.logmel_extractor.melW, melW)


# ---- GPU ----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_glue_kernels_match_torch():
    from audioldm2_amd import ops
    from audioldm2_amd.clap import sinc_resample_kernel
    g = torch.Generator().manual_seed(3)
    dev = torch.device("cuda")
    # resampler: 16 -> 48 kHz (the path's case) and a ratio with a long filter (22.05 -> 48 kHz: 147 -> 320)
    for orig, new, T in ((16000, 48000, 5003), (22050, 48000, 4099), (48000, 16000, 6001)):
        x = torch.randn(3, T, generator=g)
        k, width, down, up = sinc_resample_kernel(orig, new)
        n_out = int(math.ceil(up * T / down))
        y = ops.resample_sinc(x.to(dev), k.to(dev), down, up, width, n_out)
        assert _rel(y.cpu(), oh.resample(x, orig, new)) < 2e-6, (orig, new)
    # |STFT|^2
    spec = torch.randn(37, 2 * 513, generator=g)
    pw = ops.power_spec(spec.to(dev), 513, 516).cpu()
    assert torch.equal(pw[:, 513:], torch.zeros(37, 3))
    assert _rel(pw[:, :513], spec[:, :513] ** 2 + spec[:, 513:] ** 2) < 1e-6
    # per-column affine
    x = torch.randn(2, 50, 64, generator=g)
    sc, sh = torch.randn(64, generator=g), torch.randn(64, generator=g)
    assert _rel(ops.col_affine(x.to(dev), sc.to(dev), sh.to(dev)).cpu(), x * sc + sh) < 1e-6
    # bicubic stretch + fold + patchify: against F.interpolate + the oracle's reshape_wav2img + unfold
    for T in (1001, 501, 1024):
        x = torch.randn(2, T, 64, generator=g) * 3
        img = oh.reshape_wav2img(x[:, None], 256, 4)                               # [B, 1, 256, 256]
        want = F.unfold(img, kernel_size=4, stride=4).transpose(1, 2)              # [B, 4096, 16]
        got = ops.bicubic_patchify(x.to(dev), 256, 4).cpu()
        assert got.shape == want.shape
        assert _rel(got, want) < 2e-6, T
    # token mean, cosine
    x = torch.randn(3, 64, 1024, generator=g)
    assert _rel(ops.token_mean(x.to(dev)).cpu(), x.mean(1)) < 1e-6
    a, b = torch.randn(5, 512, generator=g), torch.randn(5, 512, generator=g)
    a[4] = 0
    got = ops.row_cosine(a.to(dev), b.to(dev)).cpu()
    assert torch.allclose(got, F.cosine_similarity(a, b, dim=-1), atol=1e-6)
    assert float(got[4]) == 0.0


def _product(prob=0.0, sampling_rate=16000):
    from audioldm2_amd.clap import CLAPAudioEmbeddingClassifierFreev2
    sd, _ = _sd()
    m = CLAPAudioEmbeddingClassifierFreev2(embed_mode="audio", unconditional_prob=prob, sampling_rate=sampling_rate,
                                           config=cases.clap_text_test_config(), audio_config=cases.htsat_test_config())
    tsd = cases_text_sd()
    m.model.load_state_dict({**sd, **tsd}, strict=False)
    return m, sd, tsd


def cases_text_sd():
    from oracle import weights
    with open(os.path.join(GOLD, "clap_text_keys.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    return weights.make_state_dict(shapes, seed=0)


@pytest.mark.gpu
def test_hip_audio_tower_matches_the_real_htsat_fixture():
    g = _fixture()
    m, sd, _ = _product()
    wav16 = cases.clap_waveform(2)
    k, width, down, up = __import__("audioldm2_amd.clap", fromlist=["x"]).sinc_resample_kernel(16000, 48000)
    from audioldm2_amd import ops
    w48 = ops.resample_sinc(wav16.cuda(), k.cuda(), down, up, width, 3 * wav16.shape[1])[:, :480000].contiguous()
    e_res = _rel(w48[:, :4096].cpu(), g["wav48_head"])
    emb_h = m.model.audio_branch({"waveform": w48})["embedding"]
    e_h = _rel(emb_h.cpu(), g["embedding"])
    ids, mask = cases.clap_text_tokens()
    m.build_unconditional_emb({"input_ids": ids[2:3].repeat(2, 1), "attention_mask": mask[2:3].repeat(2, 1)})
    out = m(wav16[:, None])                                    # forward() in "audio" mode: [bs, 1, t] -> [bs, 1, 512]
    e_e = _rel(out[:, 0].cpu(), g["emb"])
    print(f"clap audio tower vs the real HTSAT fixture: resample {e_res:.2e}  embedding {e_h:.2e}  normalised {e_e:.2e}")
    assert tuple(out.shape) == (2, 1, 512)
    assert e_res < 2e-6 and e_h < 1e-4 and e_e < 1e-4


@pytest.mark.gpu
def test_cos_similarity_matches_the_oracle_and_replays_the_unconditional_draws():
    from oracle import clap_text
    m, sd, tsd = _product()
    ids, mask = cases.clap_text_tokens()
    m.build_unconditional_emb({"input_ids": ids[2:3].repeat(2, 1), "attention_mask": mask[2:3].repeat(2, 1)})
    wav = cases.clap_waveform(3)
    tok = {"input_ids": ids, "attention_mask": mask}
    sim = m.cos_similarity(wav, tok)
    a = oh.audio_embedding(sd, wav, 16000, cases.htsat_test_config())
    t = clap_text.text_embedding(tsd, cases.clap_text_test_config(), ids, mask)
    want = oh.cos_similarity(a, t)
    assert tuple(sim.shape) == (3,)
    assert torch.allclose(sim.cpu(), want, atol=2e-5), (sim.cpu(), want)
    assert m.embed_mode == "audio"                                  # restored (encoders/modules.py:652)
    # with the reference's default probability the same host draws pick the replaced rows: 3 audio draws, then 3 text draws
    m.unconditional_prob = 0.5
    torch.manual_seed(11)
    sim2 = m.cos_similarity(wav, tok)
    torch.manual_seed(11)
    draws = [float(torch.rand(1)) < 0.5 for _ in range(6)]
    u = m.unconditional_token[0].cpu()
    a2 = torch.stack([u if draws[i] else a[i] for i in range(3)])
    t2 = torch.stack([u if draws[3 + i] else t[i] for i in range(3)])
    assert torch.allclose(sim2.cpu(), oh.cos_similarity(a2, t2), atol=2e-5)


@pytest.mark.gpu
def test_48k_input_skips_the_resampler_and_short_clips_are_stretched():
    m, sd, _ = _product(sampling_rate=48000)
    ids, mask = cases.clap_text_tokens()
    m.build_unconditional_emb({"input_ids": ids[2:3].repeat(2, 1), "attention_mask": mask[2:3].repeat(2, 1)})
    wav = oh.resample(cases.clap_waveform(1), 16000, 48000)[:, :240000]          # 5 s: 501 frames -> bicubic stretch x2
    out = m(wav)
    e = oh.htsat_embedding(sd, wav, cases.htsat_test_config())
    e = F.linear(torch.relu(F.linear(e, sd["audio_projection.0.weight"], sd["audio_projection.0.bias"])),
                 sd["audio_projection.2.weight"], sd["audio_projection.2.bias"])
    assert _rel(out[:, 0].cpu(), F.normalize(e, dim=-1)) < 1e-4


class _StubTokenizer:
    """Stands in for RobertaTokenizer.from_pretrained("roberta-base") (Hub unreachable offline): deterministic ids from the
    characters of each prompt, padded like `padding="max_length"` (to 64 here)."""

    def __call__(self, texts, padding=None, truncation=None, max_length=None, return_tensors=None):
        texts = [texts] if isinstance(texts, str) else list(texts)
        T = 64
        ids = torch.ones(len(texts), T, dtype=torch.long)
        mask = torch.zeros(len(texts), T, dtype=torch.long)
        for b, s in enumerate(texts):
            body = [3 + (ord(ch) * 7 + i) % 500 for i, ch in enumerate(s)][: T - 2]
            row = [0] + body + [2]
            ids[b, : len(row)] = torch.tensor(row)
            mask[b, : len(row)] = 1
        return {"input_ids": ids, "attention_mask": mask}


@pytest.mark.gpu
def test_generate_batch_reranks_candidates_like_the_reference():
    """ddpm.py:1554-1568 end to end: n_candidate_gen_per_text = 2 candidates per prompt through the HIP sampler, decoder and
    vocoder, ranked by the HIP CLAP towers; the chosen indices must be the ones the CPU oracle's similarities pick on the
    same candidate waveforms."""
    from audioldm2_amd.clap import CLAPAudioEmbeddingClassifierFreev2
    from audioldm2_amd.pipeline import build_model, make_batch_for_text_to_audio, seed_everything
    from oracle import clap_text
    torch.manual_seed(3)
    m = build_model(model_name="audioldm2-full").cuda()
    assert m.clap is None
    with pytest.raises(NotImplementedError):          # fails BEFORE sampling when built without the re-ranker
        m.generate_batch(make_batch_for_text_to_audio("x", batchsize=1), ddim_steps=2, n_gen=2, duration=10)
    if torch.is_tensor(m.scale_factor):
        m.scale_factor.fill_(0.75)
    clap = CLAPAudioEmbeddingClassifierFreev2(embed_mode="audio", unconditional_prob=0.0, sampling_rate=16000,
                                              config=cases.clap_text_test_config(), audio_config=cases.htsat_test_config())
    sd, _ = _sd()
    tsd = cases_text_sd()
    clap.model.load_state_dict({**sd, **tsd}, strict=False)
    clap.tokenize = _StubTokenizer()
    m.clap = clap
    seen = {}
    orig = clap.cos_similarity

    def spy(waveform, text):
        seen["waveform"], seen["text"] = waveform.clone(), list(text)
        return orig(waveform, text)
    clap.cos_similarity = spy
    B0 = 2
    batch = make_batch_for_text_to_audio("a dog barking in the rain", batchsize=B0)
    batch["text"][1] = "a slow piano melody"
    seed_everything(7)
    m.latent_t_size = 256
    wav = m.generate_batch(batch, unconditional_guidance_scale=3.5, ddim_steps=2, n_gen=2, duration=10)
    assert wav.shape == (B0, 1, 163872) and np.isfinite(wav).all()
    cand = seen["waveform"]
    assert tuple(cand.shape) == (2 * B0, 163872) and seen["text"] == batch["text"] * 2      # ddpm.py:1515: text repeated n_gen x
    tok = _StubTokenizer()(seen["text"])
    a = oh.audio_embedding(sd, cand.float(), 16000, cases.htsat_test_config())
    t = clap_text.text_embedding(tsd, cases.clap_text_test_config(), tok["input_ids"], tok["attention_mask"])
    want = oh.cos_similarity(a, t)
    assert torch.allclose(m.last_similarity, want, atol=5e-5), (m.last_similarity, want)
    best = [i + int(torch.argmax(want[i::B0])) * B0 for i in range(B0)]
    assert m.last_best_index == best
    assert np.array_equal(wav[:, 0], cand.numpy()[best])


@pytest.mark.gpu
def test_read_wav_file_resamples_like_the_reference(tmp_path):
    """utilities/audio/tools.py:28-40: a .wav at another rate is resampled to 16 kHz (torchaudio.functional.resample in the
    reference; the restated polyphase sinc filter on the GPU here), then normalised / padded / scaled like the reference."""
    from scipy.io import wavfile
    from audioldm2_amd.pipeline import normalize_wav, pad_wav, read_wav_file
    sr = 44100
    t = np.arange(int(1.7 * sr)) / sr
    x = (0.4 * np.sin(2 * np.pi * 440.0 * t) + 0.2 * np.sin(2 * np.pi * 3000.0 * t + 0.5)).astype(np.float32)
    path = str(tmp_path / "tone44k.wav")
    wavfile.write(path, sr, x)
    got = read_wav_file(path, 163840)
    ref16 = oh.resample(torch.from_numpy(x)[None], sr, 16000)[0].numpy()
    want = pad_wav(normalize_wav(ref16)[None, ...], 163840)
    want = 0.5 * want / np.max(np.abs(want))
    assert got.shape == want.shape == (1, 163840)
    assert np.abs(got - want).max() < 2e-6
    # and the 16 kHz signal really is the 440 Hz + 3 kHz pair (the filter kept both, at the new rate)
    n = ref16.shape[0]
    t16 = np.arange(n) / 16000.0
    ideal = 0.4 * np.sin(2 * np.pi * 440.0 * t16) + 0.2 * np.sin(2 * np.pi * 3000.0 * t16 + 0.5)
    assert np.abs(ref16[400:-400] - ideal[400:-400]).max() < 3e-3
