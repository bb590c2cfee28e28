"""CPU tests: the oracle restatement (oracle/*.py) against the fixtures that oracle/make_golden.py
produced by running the REAL reference classes (tests/golden/*.npz).  This is what pins the oracle.
Tolerance: the oracle performs the same ATen ops in the same order, so agreement is at fp32
round-off of thread-count-dependent reductions: max|err| <= 2e-5 * max|ref| (usually exactly 0).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cases, weights
from oracle.ddim import ddim_tables, make_schedule_buffers
from oracle.unet import unet_forward
from oracle.vae import hifigan_forward, vae_decode, vae_encode_moments

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-5


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def rel(a, b):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _shapes(fname, key):
    import json
    with open(os.path.join(GOLD, fname)) as f:
        return {k: tuple(v) for k, v in json.load(f)[key].items()}


@pytest.mark.parametrize("name,cfg,B,H,W,t5", [
    ("unet_tiny", cases.UNET_TINY, 2, 16, 8, 12),
    ("unet_large_tiny", cases.UNET_LARGE_TINY, 2, 16, 8, 12),
    ("unet_film_tiny", cases.UNET_FILM_TINY, 2, 8, 16, 12),
    ("unet_full", cases.UNET_FULL, 1, 256, 16, 32),
])
def test_unet_oracle_matches_reference_fixture(name, cfg, B, H, W, t5):
    sd = weights.make_state_dict(_shapes("unet_statedict_keys.json", name), seed=0)
    x, t, ctxs, masks, y = cases.unet_inputs(cfg, B, H, W, t5)
    out = unet_forward(sd, cfg, x, t, ctxs, masks, y=y)
    assert rel(out, gold(name)["out"]) < TOL


@pytest.mark.parametrize("name,dd,shapes", [
    ("vae16k", cases.DDCONFIG_16K, [(2, 8, 32, 16), (1, 8, 256, 16)]),
    ("vae48k", cases.DDCONFIG_48K, [(1, 16, 16, 32)]),
])
def test_vae_oracle_matches_reference_fixture(name, dd, shapes):
    sd = weights.make_state_dict(_shapes("vae_statedict_keys.json", name), seed=0)
    g = gold(name)
    for i, shp in enumerate(shapes):
        mel = vae_decode(sd, dd, cases.latent_input(*shp, seed=i))
        assert rel(mel, g[f"mel{i}"]) < TOL
    f = 2 ** (len(dd["ch_mult"]) - 1)
    x = cases.mel_input(1, dd["mel_bins"], 16 * f, seed=5).permute(0, 2, 1)[:, None]
    assert rel(vae_encode_moments(sd, dd, x), g["moments"]) < TOL


@pytest.mark.parametrize("name,hc,Ts", [("hifigan16k", cases.HIFIGAN_16K, [48, 1024]),
                                        ("hifigan48k", cases.HIFIGAN_48K, [24])])
def test_hifigan_oracle_matches_reference_fixture(name, hc, Ts):
    shapes = {k: v for k, v in _shapes("vae_statedict_keys.json", "vae16k" if "16k" in name else "vae48k").items()
              if k.startswith("vocoder.")}
    sd = weights.make_state_dict({k[len("vocoder."):]: v for k, v in shapes.items()}, seed=0)
    g = gold(name)
    for i, T in enumerate(Ts):
        w = hifigan_forward(sd, hc, cases.mel_input(1, hc["num_mels"], T, seed=i))
        assert rel(w, g[f"wave{i}"]) < TOL


def test_ddim_tables_match_reference_exactly():
    """ddim.py:33-91 tables, including the reference's mixed float32/float64 arithmetic: bit-exact."""
    g = gold("ddim_tables")
    ac = make_schedule_buffers(1000, 0.0015, 0.0195)["alphas_cumprod"]
    for S, eta in [(200, 1.0), (50, 0.0), (5, 1.0)]:
        ts, coef = ddim_tables(ac, S, eta)
        assert np.array_equal(ts, g[f"ts_{S}"])
        a = torch.from_numpy(g[f"alphas_{S}"])
        ap = torch.from_numpy(g[f"alphas_prev_{S}"])
        sg = torch.from_numpy(g[f"sigmas_{S}"])
        som = torch.from_numpy(g[f"som_{S}"])
        for i in range(S):  # coefficient rows as p_sample_ddim forms them (ddim.py:330-353)
            a_t = torch.full((1,), float(a[i]))
            a_prev = torch.full((1,), float(ap[i]))
            s_t = torch.full((1,), float(sg[i]))
            ref = torch.cat([torch.full((1,), float(som[i])), a_t.sqrt(), (1.0 - a_prev - s_t ** 2).sqrt(),
                             a_prev.sqrt(), s_t])
            assert torch.equal(coef[i], ref), (S, i)
    # SURVEY.md §8(a3) check values
    _, c200 = ddim_tables(ac, 200, 1.0)
    assert abs(float(c200[0, 4]) - 0.027432) < 1e-6 and abs(float(c200[199, 4]) - 0.305154) < 1e-6


def test_stft_oracle_matches_reference_fixture():
    """Reference TacotronSTFT (with the oracle's mel filterbank injected for the absent librosa)."""
    from oracle import stft as ostft
    g = gold("stft16k")
    x = cases.wave_input(2, 16000, seed=0)
    mel, mag, phase, energy = ostft.mel_spectrogram(x)
    assert rel(mag, g["mag"]) < TOL and rel(mel, g["mel"]) < 1e-4 and rel(energy, g["energy"]) < TOL
    assert np.array_equal(ostft.stft_forward_basis(1024, 1024)[:8], g["basis_head"])
    # 440 Hz known answer: peak bin of the sine row = round(440 / (16000/1024)) = 28
    assert int(mag[0, :, 50].argmax()) == 28
    # filterbank sanity (librosa 0.9.2 defaults): 64 slaney filters, unit-area-ish normalisation
    fb = ostft.mel_filterbank(16000, 1024, 64, 0, 8000)
    assert fb.shape == (64, 513) and fb.min() >= 0 and (fb.sum(1) > 0).all()


@pytest.mark.timeout(900)
def test_e2e_oracle_matches_reference_generate_batch_5step():
    """Whole path (RNG contract, conditioner routing, CFG, DDIM, VAE decode, vocoder) vs the real
    LatentDiffusion.generate_batch fixture (B=2, 5 DDIM steps, CFG 3.5, seed 42)."""
    from oracle.pipeline import OracleLatentDiffusion
    g = gold("e2e_full_5step_b2")
    o = OracleLatentDiffusion()
    torch.manual_seed(cases.E2E_SEED)
    out = o.generate_batch(cases.e2e_batch(2), unconditional_guidance_scale=3.5, ddim_steps=5)
    assert rel(out["latent"], g["latent"]) < 1e-4
    assert rel(out["mel"], g["mel"]) < 1e-4
    assert rel(out["wave"], g["wave"]) < 1e-4


@pytest.mark.timeout(900)
def test_oracle_two_jobs_in_one_process_match_the_real_reference_second_call_included():
    """The draw nobody asked for (ddpm.py:850-855, 916-917): from its SECOND `get_input` on, a reference `LatentDiffusion` draws one
    `torch.rand(1)` per job (make_decision(unconditional_prob_cfg = 0.0): always "no", but the generator moves), so the same seed gives a
    different clip on the second call.  `tests/golden/parity_script_ref_s5b2_s20b1.npz` was written by the REAL reference running TWO jobs
    in one process on the deterministic checkpoint of `tools/parity_on_checkpoint.py --make-random-ckpt` (oracle.weights seed 3): 5 steps x
    2 prompts (first call), then 20 steps x 1 prompt (second call).  The oracle — and the product, in the GPU suite — must reproduce
    both; without the second call's extra draw job 2 is a different clip (relative error ~1.4, the bug this test pins)."""
    from oracle import weights
    from oracle.pipeline import OracleLatentDiffusion, hot_path_shapes
    g = gold("parity_script_ref_s5b2_s20b1")
    o = OracleLatentDiffusion(sd=weights.make_state_dict(hot_path_shapes(), seed=3))
    torch.manual_seed(cases.E2E_SEED)
    a = o.generate_batch(cases.e2e_batch(2), unconditional_guidance_scale=3.5, ddim_steps=5)
    assert rel(a["latent"], g["s5_b2_latent"]) < 1e-4 and rel(a["wave"], g["s5_b2_wave"]) < 1e-4
    assert o.conditional_dry_run_finished
    torch.manual_seed(cases.E2E_SEED)
    b = o.generate_batch(cases.e2e_batch(1), unconditional_guidance_scale=3.5, ddim_steps=20)
    assert rel(b["latent"], g["s20_b1_latent"]) < 1e-4 and rel(b["mel"], g["s20_b1_mel"]) < 1e-4
    assert rel(b["wave"], g["s20_b1_wave"]) < 1e-4


def test_e2e_oracle_matches_reference_generate_batch_masked():
    """Inpainting / super-resolution path vs the real LatentDiffusion.generate_batch_masked fixture
    (B=1, 4 DDIM steps, CFG 2.5, seed 42): VAE encode + posterior draw, mask, q_sample blend order."""
    from oracle.pipeline import OracleLatentDiffusion
    g = gold("e2e_masked_4step_b1")
    o = OracleLatentDiffusion()
    torch.manual_seed(cases.E2E_SEED)
    out = o.generate_batch_masked(cases.e2e_masked_batch(1), unconditional_guidance_scale=2.5, ddim_steps=4)
    assert rel(out["x0"], g["x0"]) < 1e-4
    assert torch.equal(out["mask"], torch.from_numpy(g["mask"]))
    assert rel(out["latent"], g["latent"]) < 1e-4
    assert rel(out["wave"], g["wave"]) < 1e-4


def test_ancestral_oracle_matches_reference_sample():
    """Ancestral DDPM sampler vs the real LatentDiffusion.sample(timesteps=4) fixture."""
    from oracle.pipeline import OracleLatentDiffusion
    g = gold("ancestral_4step_b1")
    o = OracleLatentDiffusion()
    torch.manual_seed(cases.E2E_SEED)
    z, first = o.sample_ancestral(cases.e2e_batch(1), 4)
    assert rel(first, g["first"]) < 1e-4
    assert rel(z, g["latent"]) < 1e-4


def test_e2e_oracle_matches_reference_generate_batch_48k():
    """BASELINE config 3 (audioldm_48k) vs the real reference's generate_batch fixture (B=1, 2 DDIM steps,
    CFG 3.5): FiLM routing (y), 16-channel 128x32 latent, 4-level VAE decode, 48 kHz vocoder."""
    from oracle.pipeline import oracle_48k
    g = gold("e2e_48k_2step_b1")
    o = oracle_48k()
    torch.manual_seed(cases.E2E_SEED)
    out = o.generate_batch(cases.e2e_batch_48k(1), unconditional_guidance_scale=3.5, ddim_steps=2)
    assert out["wave"].shape[-1] == int(g["wave_len"])
    assert rel(out["latent"], g["latent"]) < 1e-4
    assert rel(out["wave"][..., :32768], g["wave_head"]) < 1e-4
    assert rel(out["wave"][..., ::16], g["wave_dec"]) < 1e-4


@pytest.mark.parametrize("model_name,fixture", [("audioldm2-speech-gigaspeech", "e2e_speech_2step_b1"),
                                                ("audioldm2-full-large-1150k", "e2e_large_2step_b1")])
def test_e2e_oracle_matches_reference_generate_batch_speech_and_large(model_name, fixture):
    """BASELINE configs 4 / 5 vs the real reference's generate_batch fixtures (B=1, 2 DDIM steps, CFG 3.5): one
    512-token context (speech) and three context slots with transformer depth 2 (large)."""
    from oracle.pipeline import oracle_named
    g = gold(fixture)
    o = oracle_named(model_name)
    torch.manual_seed(cases.E2E_SEED)
    out = o.generate_batch(cases.e2e_batch(1), unconditional_guidance_scale=3.5, ddim_steps=2)
    assert out["wave"].shape[-1] == int(g["wave_len"])
    assert rel(out["latent"], g["latent"]) < 1e-4
    assert rel(out["wave"][..., :32768], g["wave_head"]) < 1e-4
    assert rel(out["wave"][..., ::16], g["wave_dec"]) < 1e-4


@pytest.mark.parametrize("sr,n_fft,n_mels,fmin,fmax", [(16000, 1024, 64, 0, 8000), (48000, 2048, 256, 20, 24000)])
def test_mel_filterbank_agrees_with_an_independent_slaney_implementation(sr, n_fft, n_mels, fmin, fmax):
    """librosa==0.9.2 (the reference's mel basis, stft.py:145-147) is not installed, so oracle/stft.py restates the
    published Slaney definition; transformers.audio_utils.mel_filter_bank is a second, independent restatement of the
    same definition (norm="slaney", mel_scale="slaney") and must give the same basis for both shipped STFT configs."""
    au = pytest.importorskip("transformers.audio_utils")
    from oracle import stft as ost
    ours = np.asarray(ost.mel_filterbank(sr, n_fft, n_mels, fmin, fmax), dtype=np.float64)
    theirs = au.mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=fmin,
                                max_frequency=fmax, sampling_rate=sr, norm="slaney", mel_scale="slaney").T
    assert ours.shape == theirs.shape == (n_mels, n_fft // 2 + 1)
    assert np.abs(ours - theirs).max() < 1e-7 * np.abs(theirs).max() + 1e-9


# ---- §8(f) rank 1: the AudioMAE-token sequence generator (GPT-2 autoregressive loop) --------------------------------
def _seqgen_sd(fixture):
    with open(os.path.join(GOLD, fixture + "_keys.json")) as f:
        return weights.make_state_dict({k: tuple(v) for k, v in json.load(f).items()}, seed=0)


@pytest.mark.parametrize("fixture,cfg,T", [("seqgen_full_8step_b2", cases.SEQGEN_FULL, 20),
                                           ("seqgen_speech_24step_b2", cases.SEQGEN_SPEECH, 40)])
def test_sequence_generator_oracle_matches_reference_generate(fixture, cfg, T):
    """oracle/seqgen.py vs the REAL Sequence2AudioMAE.generate (sequence_input.py:294-325, transformers GPT2Model) on the
    committed fixture: the reference's full re-forward order and the key/value-cached order both reproduce it."""
    from oracle import seqgen
    sd = _seqgen_sd(fixture)
    want = torch.from_numpy(gold(fixture)["out"])
    x, mask = seqgen.input_sequence_and_mask(sd, cases.seqgen_cond(cfg, 2, T), cfg["keys"], cfg["steps"])
    assert x.shape[1] == T + 2 + 1 + 2  # [sos, clap, eos] + [sos, T tokens, eos]
    assert rel(seqgen.generate_full(sd, x, mask, cfg["steps"]), want) < 1e-5
    assert rel(seqgen.generate_cached(sd, x, mask, cfg["steps"]), want) < 1e-5


def test_sequence_generator_cached_decode_equals_full_reforward_over_many_steps():
    """The size-independent property an accelerated decode rests on: with a causal model the cached evaluation order
    is the same function as the reference's O(n^2) re-forward loop, here over 96 generated positions."""
    from oracle import seqgen
    sd = _seqgen_sd("seqgen_speech_24step_b2")
    cfg = cases.SEQGEN_SPEECH
    x, mask = seqgen.input_sequence_and_mask(sd, cases.seqgen_cond(cfg, 1, 12, seed=9), cfg["keys"], 96)
    a = seqgen.generate_full(sd, x, mask, 96)
    b = seqgen.generate_cached(sd, x, mask, 96)
    assert a.shape == (1, 96, 768) and rel(b, a) < 1e-5
