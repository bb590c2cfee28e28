"""The oracle restatement against the LIVE reference classes (imported from /root/reference through
oracle/refimport.py) on inputs that are NOT in the committed fixtures.  Runs only where the reference
checkout exists (the build container); skipped on the GPU box.  CPU only."""
import pytest
import torch

from oracle import cases, refimport, weights
from oracle.unet import unet_forward
from oracle.vae import hifigan_forward, vae_decode, vae_encode_moments

pytestmark = pytest.mark.skipif(not refimport.available(), reason="reference checkout not present")


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _load_det(module, seed):
    sd = weights.make_state_dict(weights.shapes_of(module), seed=seed)
    module.load_state_dict(sd)
    return sd


@pytest.mark.parametrize("cfg,B,H,W,t5,seed", [(cases.UNET_TINY, 3, 24, 8, 5, 11), (cases.UNET_LARGE_TINY, 1, 16, 16, 20, 12),
                                              (cases.UNET_FILM_TINY, 2, 16, 8, 0, 13)])
def test_unet_oracle_equals_live_reference(cfg, B, H, W, t5, seed):
    """openaimodel.UNetModel.forward (openaimodel.py:837-885) on fresh weights / shapes / seeds."""
    ref = refimport.unet_cls()(**cfg).eval()
    sd = _load_det(ref, seed)
    x, t, ctxs, masks, y = cases.unet_inputs(cfg, B, H, W, max(t5, 1), seed=seed)
    with torch.no_grad():
        want = ref(x, t, y=y, context_list=ctxs, context_attn_mask_list=masks)
        got = unet_forward(sd, cfg, x, t, ctxs, masks, y=y)
    assert rel(got, want) < 1e-5


def test_vae_oracle_equals_live_reference():
    """AutoencoderKL.decode / .encode (autoencoder.py:103-117) for the 16 kHz ddconfig."""
    refimport.install()
    from audioldm2.latent_encoder.autoencoder import AutoencoderKL
    dd = cases.DDCONFIG_16K
    ae = AutoencoderKL(ddconfig=dd, embed_dim=dd["z_channels"], image_key="fbank").eval()
    sd = _load_det(ae, 21)
    z = cases.latent_input(1, 8, 24, 16, seed=21)
    x = cases.mel_input(1, 64, 64, seed=22).permute(0, 2, 1)[:, None]
    with torch.no_grad():
        assert rel(vae_decode(sd, dd, z), ae.decode(z)) < 1e-5
        assert rel(vae_encode_moments(sd, dd, x), ae.encode(x).parameters) < 1e-5


def test_hifigan_oracle_equals_live_reference():
    """hifigan.Generator.forward (hifigan/models.py:149-165), 16 kHz config."""
    g = refimport.hifigan_generator(dict(cases.HIFIGAN_16K))
    sd = _load_det(g, 31)
    mel = cases.mel_input(2, 64, 20, seed=31)
    with torch.no_grad():
        assert rel(hifigan_forward(sd, cases.HIFIGAN_16K, mel), g(mel)) < 1e-5


def test_audio_front_end_helpers_equal_live_reference():
    """pad_wav / normalize_wav / _pad_spec of the inpainting front-end (utilities/audio/tools.py:9-25,71-84)
    — host helpers of audioldm2_amd.pipeline — against the reference's own functions."""
    import numpy as np
    refimport.install()
    import audioldm2.utilities.audio.tools as rt
    from audioldm2_amd import pipeline as P
    rng = np.random.default_rng(0)
    w = rng.standard_normal(5000).astype(np.float32) * 0.3 + 0.1
    assert np.array_equal(P.normalize_wav(w), rt.normalize_wav(w))
    for seg in (None, 5000, 8000):
        assert np.array_equal(P.pad_wav(w[None], seg), rt.pad_wav(w[None], seg))
    fb = torch.randn(1000, 65)
    for tl in (1024, 900):
        assert torch.equal(P._pad_spec(fb, tl), rt._pad_spec(fb, tl))


def test_sequence_generator_oracle_equals_live_reference():
    """§8(f) rank 1: Sequence2AudioMAE.generate (sequence_input.py:294-325) on conditioning that is not in the committed
    fixtures (other batch, lengths, seed), full re-forward and key/value-cached restatements."""
    from oracle import seqgen
    cfg = dict(cases.SEQGEN_FULL, steps=5)
    m = refimport.sequence_generator(cfg["steps"], cfg["keys"], cfg["dims"])
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if k != "model.wte.weight"}
    sd = weights.make_state_dict(shapes, seed=3)
    m.load_state_dict(sd, strict=False)
    cond = cases.seqgen_cond(cfg, 3, 9, seed=21)
    with torch.no_grad():
        want, _ = m.generate(None, cond_dict=cond)
    x, mask = seqgen.input_sequence_and_mask(sd, cond, cfg["keys"], cfg["steps"])
    assert rel(seqgen.generate_full(sd, x, mask, cfg["steps"]), want) < 1e-5
    assert rel(seqgen.generate_cached(sd, x, mask, cfg["steps"]), want) < 1e-5


def test_htsat_oracle_equals_live_reference():
    """§8(f) rank 4: HTSAT_Swin_Transformer.forward(...)["embedding"] (clap/open_clip/htsat.py:1092-1127) on audio that is not
    in the committed fixture — other weights, one sample, a 6 s clip (601 frames: the bicubic stretch of reshape_wav2img runs
    at a different ratio) — with depths (1, 2, 1, 2): a stage that ends on an unshifted block and one with a shifted one."""
    from oracle import htsat as oh
    hc = dict(cases.htsat_test_config(), depths=(1, 2, 1, 2))
    ac = dict(oh.AUDIO_CFG)
    m = refimport.htsat_swin_transformer(hc, ac)
    skip = ("relative_position_index", "attn_mask", "num_batches_tracked", "tscam_conv", "head.")
    shapes = {"audio_branch." + k: tuple(v.shape) for k, v in m.state_dict().items() if not any(t in k for t in skip)}
    sd = cases.htsat_state_dict(shapes, seed=5)
    missing = m.load_state_dict({k[len("audio_branch."):]: v for k, v in sd.items()}, strict=False)
    assert all(any(t in k for t in skip) for k in missing.missing_keys) and not missing.unexpected_keys
    wav48 = oh.resample(cases.clap_waveform(1, seed=33), 16000, 48000)[:, :288000]
    with torch.no_grad():
        want = m({"waveform": wav48}, device="cpu")["embedding"]
    assert rel(oh.htsat_embedding(sd, wav48, hc, ac), want) < 1e-5


def test_phoneme_and_t5_oracles_equal_live_reference():
    """§8(f) ranks 1 (second half) and 2: the REAL PhonemeEncoder and FlanT5HiddenState on inputs and weights that are not in
    the committed fixtures."""
    from oracle import phoneme as oph
    from oracle import t5 as ot5
    # VITS phoneme encoder: other weights, other ids / lengths
    m = refimport.phoneme_encoder(**cases.PHONEME)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = cases.phoneme_state_dict(shapes, seed=4)
    m.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(17)
    idx = torch.randint(1, cases.PHONEME["vocabs_size"], (3, cases.PHONEME["pad_length"]), generator=g)
    for b, n in enumerate((310, 57, 1)):
        idx[b, n:] = cases.PHONEME["pad_token_id"]
    with torch.no_grad():
        want, wmask = m(idx)
    emb, mask = oph.phoneme_encoder_forward(sd, idx, cases.PHONEME["pad_token_id"])
    assert rel(emb, want) < 1e-5 and torch.equal(mask, wmask)
    # FLAN-T5 encoder: other weights, other token batch (lengths 9, 17, 2 padded to 17)
    cfg = cases.t5_test_config()
    ids = torch.randint(3, cfg["vocab_size"], (3, 17), generator=g)
    am = torch.zeros(3, 17, dtype=torch.long)
    for b, n in enumerate((9, 17, 2)):
        ids[b, n - 1] = 1
        ids[b, n:] = 0
        am[b, :n] = 1
    ref = refimport.flan_t5_hidden_state(cfg, lambda prompt: (ids, am))
    tshapes = {k: tuple(v.shape) for k, v in ref.model.state_dict().items()}
    tsd = cases.t5_state_dict(tshapes, seed=6)
    ref.model.load_state_dict(tsd, strict=True)
    with torch.no_grad():
        hs, ram = ref(["x", "y", "z"])
    h, m2 = ot5.encode_tokens(tsd, cfg, ids, am)
    assert rel(h, hs) < 1e-5 and torch.equal(m2.float(), ram.float())
