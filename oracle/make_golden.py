"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference with the
import stubs of oracle/refimport.py) on the seeded cases of oracle/cases.py.  TEST INFRA ONLY; run
in the build container:  python -m oracle.make_golden [all|unet|vae|hifigan|ddim|stft|e2e5|masked|ancestral|e2e200]

The reference ships no golden vectors; these fixtures are what pins both the oracle
restatement (tests/test_oracle.py, CPU) and the HIP product (tests/test_*_gpu.py) to the reference.
Weights are the deterministic name-keyed tensors of oracle/weights.py loaded through the reference
modules' own load_state_dict.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from oracle import cases, refimport, weights

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"  wrote {path} ({os.path.getsize(path)/1024:.0f} KiB)")


def rms64(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt((a ** 2).mean()))


def between_sample_rms(ld, wav, latent):
    """What a waveform tolerance has to be read against (VERDICT r2 "weak" #1): the rms DIFFERENCE between the waveforms of two
    unrelated samples.  B >= 2: samples 0 and 1 of the batch; B = 1: the sample against the decode + vocode of a second,
    unrelated latent of the same statistics (seeded)."""
    if wav.shape[0] >= 2:
        return rms64(wav[0].astype(np.float64) - wav[1].astype(np.float64))
    g = torch.Generator().manual_seed(9)
    z2 = torch.randn(latent.shape, generator=g) * latent.std() + latent.mean()
    mel2 = ld.decode_first_stage(z2)
    wav2 = ld.mel_spectrogram_to_waveform(mel2, savepath="", bs=None, name="x", save=False)
    return rms64(wav[0].astype(np.float64) - wav2[0].astype(np.float64))


def load_det(module, seed=0):
    sd = weights.make_state_dict(weights.shapes_of(module), seed=seed)
    module.load_state_dict(sd)
    return sd


def gen_unet():
    U = refimport.unet_cls()
    keys = {}
    for name, cfg, B, H, W, t5 in [("unet_tiny", cases.UNET_TINY, 2, 16, 8, 12),
                                   ("unet_large_tiny", cases.UNET_LARGE_TINY, 2, 16, 8, 12),
                                   ("unet_film_tiny", cases.UNET_FILM_TINY, 2, 8, 16, 12),
                                   ("unet_full", cases.UNET_FULL, 1, 256, 16, 32)]:
        torch.manual_seed(0)
        ref = U(**cfg).eval()
        load_det(ref)
        keys[name] = {k: list(v) for k, v in weights.shapes_of(ref).items()}
        x, t, ctxs, masks, y = cases.unet_inputs(cfg, B, H, W, t5)
        t0 = time.time()
        with torch.no_grad():
            out = ref(x, t, y=y, context_list=ctxs, context_attn_mask_list=masks)
        print(f"{name}: ref forward {time.time()-t0:.1f}s out std {out.std():.4f}")
        save(name, out=out)
    with open(os.path.join(OUT, "unet_statedict_keys.json"), "w") as f:
        json.dump(keys, f)


def _ref_autoencoder(dd):
    refimport.install()
    from audioldm2.latent_encoder.autoencoder import AutoencoderKL
    return AutoencoderKL(ddconfig=dd, embed_dim=dd["z_channels"], image_key="fbank").eval()


def gen_vae():
    keys = {}
    for name, dd, shapes in [("vae16k", cases.DDCONFIG_16K, [(2, 8, 32, 16), (1, 8, 256, 16)]),
                             ("vae48k", cases.DDCONFIG_48K, [(1, 16, 16, 32)])]:
        ae = _ref_autoencoder(dd)
        load_det(ae)
        keys[name] = {k: list(v) for k, v in weights.shapes_of(ae).items()}
        arrs = {}
        for i, shp in enumerate(shapes):
            z = cases.latent_input(*shp, seed=i)
            t0 = time.time()
            with torch.no_grad():
                mel = ae.decode(z)
            print(f"{name} decode {shp}: {time.time()-t0:.1f}s  mel std {mel.std():.3f} mean {mel.mean():.3f}")
            arrs[f"mel{i}"] = mel
        # encoder: moments of a random mel (posterior mean/logvar before sampling)
        f = 2 ** (len(dd["ch_mult"]) - 1)
        x = cases.mel_input(1, dd["mel_bins"], 16 * f, seed=5).permute(0, 2, 1)[:, None]  # [1,1,T,F]
        with torch.no_grad():
            post = ae.encode(x)
        arrs["moments"] = post.parameters
        save(name, **arrs)
    with open(os.path.join(OUT, "vae_statedict_keys.json"), "w") as fjs:
        json.dump(keys, fjs)


def gen_hifigan():
    for name, hc, Ts in [("hifigan16k", cases.HIFIGAN_16K, [48, 1024]), ("hifigan48k", cases.HIFIGAN_48K, [24])]:
        g = refimport.hifigan_generator(dict(hc))
        load_det(g)
        arrs = {}
        for i, T in enumerate(Ts):
            mel = cases.mel_input(1, hc["num_mels"], T, seed=i)
            t0 = time.time()
            with torch.no_grad():
                w = g(mel)
            print(f"{name} T={T}: {time.time()-t0:.1f}s wave {tuple(w.shape)} rms {w.pow(2).mean().sqrt():.4f} absmax {w.abs().max():.3f}")
            arrs[f"wave{i}"] = w
        save(name, **arrs)


def gen_ddim():
    D = refimport.ddim_sampler_cls()
    from oracle.ddim import make_schedule_buffers

    class Shim:
        pass
    buf = make_schedule_buffers(1000, 0.0015, 0.0195)
    m = Shim()
    m.num_timesteps = 1000
    m.betas, m.alphas_cumprod, m.alphas_cumprod_prev = buf["betas"], buf["alphas_cumprod"], buf["alphas_cumprod_prev"]
    m.device = torch.device("cpu")
    arrs = {}
    for S, eta in [(200, 1.0), (50, 0.0), (5, 1.0)]:
        s = D(m, device=torch.device("cpu"))
        s.make_schedule(S, ddim_eta=eta, verbose=False)
        arrs[f"ts_{S}"] = s.ddim_timesteps
        arrs[f"alphas_{S}"] = s.ddim_alphas
        arrs[f"alphas_prev_{S}"] = s.ddim_alphas_prev
        arrs[f"sigmas_{S}"] = s.ddim_sigmas
        arrs[f"som_{S}"] = s.ddim_sqrt_one_minus_alphas
    save("ddim_tables", **arrs)


def gen_stft():
    """The reference STFT needs librosa (absent).  pad_center is the identity for win == n_fft (both
    AudioLDM2 configs) and the mel filterbank is the oracle's restatement of librosa 0.9.2 (parity
    UNPINNED for the filterbank); everything else below is the reference's own code running."""
    refimport.install()
    from oracle import stft as ostft
    lib = sys.modules.get("librosa") or __import__("librosa")
    sys.modules["librosa.util"].pad_center = lambda w, size, **k: w if len(w) == size else np.pad(
        w, ((size - len(w)) // 2, size - len(w) - (size - len(w)) // 2))
    sys.modules["librosa.util"].tiny = lambda x: np.finfo(np.float32).tiny
    sys.modules["librosa.filters"].mel = lambda sr, n_fft, n_mels, fmin, fmax: ostft.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    for mod in ("audioldm2.utilities.audio.stft", "audioldm2.utilities.audio.audio_processing"):
        sys.modules.pop(mod, None)
    from audioldm2.utilities.audio.stft import TacotronSTFT
    st = TacotronSTFT(1024, 160, 1024, 64, 16000, 0, 8000)
    x = cases.wave_input(2, 16000, seed=0)
    mel, mag, phase, energy = st.mel_spectrogram(x)
    print("stft: mel", tuple(mel.shape), "mag max", float(mag.max()))
    save("stft16k", mel=mel, mag=mag, energy=energy, mel_basis=st.mel_basis, basis_head=st.stft_fn.forward_basis[:8, 0, :])


def _ref_latent_diffusion():
    refimport.install()
    import audioldm2.utils as ru
    from audioldm2.latent_diffusion.models.ddpm import LatentDiffusion
    P = ru.default_audioldm_config("audioldm2-full")["model"]["params"]
    P["cond_stage_config"] = cases.e2e_cond_config("cpu")
    P["device"] = "cpu"
    torch.manual_seed(0)
    ld = LatentDiffusion(**P).eval()
    sd = ld.state_dict()
    hot = {k: tuple(v.shape) for k, v in sd.items()
           if k.startswith("model.diffusion_model.") or k.startswith("first_stage_model.")}
    new = weights.make_state_dict(hot, seed=0)
    new["scale_factor"] = torch.tensor(cases.SCALE_FACTOR)
    missing, unexpected = ld.load_state_dict(new, strict=False)
    assert not unexpected
    with open(os.path.join(OUT, "e2e_statedict_keys.json"), "w") as f:
        json.dump({k: list(v) for k, v in hot.items()}, f)
    return ld


def gen_e2e(steps: int, B: int, name: str, decimate: bool = False):
    ld = _ref_latent_diffusion()
    ld.latent_t_size = 256
    rec = {}
    orig_decode = ld.decode_first_stage

    def decode_hook(z):
        rec["latent"] = z.clone()
        mel = orig_decode(z)
        rec["mel"] = mel.clone()
        return mel
    ld.decode_first_stage = decode_hook
    # seed_everything(42) of pipeline.py:20-31
    import random
    random.seed(cases.E2E_SEED)
    np.random.seed(cases.E2E_SEED)
    torch.manual_seed(cases.E2E_SEED)
    t0 = time.time()
    wav = ld.generate_batch(cases.e2e_batch(B), unconditional_guidance_scale=3.5, ddim_steps=steps, n_gen=1,
                            duration=10)
    dt = time.time() - t0
    print(f"{name}: reference generate_batch B={B} steps={steps}: {dt:.1f}s  wave {wav.shape} rms {np.sqrt((wav**2).mean()):.4f}"
          f" absmax {np.abs(wav).max():.3f}  latent std {rec['latent'].std():.3f}")
    ld.decode_first_stage = orig_decode
    btw = between_sample_rms(ld, wav, rec["latent"])
    print(f"{name}: between-sample wave rms {btw:.4f} = {btw / rms64(wav):.2f} x wave rms")
    if decimate:   # large batches: head + every 16th sample of the waveform, no mel
        save(name, latent=rec["latent"], wave_head=wav[..., :32768], wave_dec=wav[..., ::16], wave_len=np.int64(wav.shape[-1]),
             wave_rms=np.float64(rms64(wav)), wave_between_rms=np.float64(btw))
    else:
        save(name, latent=rec["latent"], mel=rec["mel"], wave=wav, seconds=np.float32(dt), threads=np.int32(torch.get_num_threads()),
             wave_between_rms=np.float64(btw))


def _ref_latent_diffusion_named(model_name: str, keys_json: str):
    """Real reference LatentDiffusion for any config name, synthetic conditioners under the reference's
    cond keys, deterministic hot-path weights."""
    refimport.install()
    import audioldm2.utils as ru
    from audioldm2.latent_diffusion.models.ddpm import LatentDiffusion
    from audioldm2_amd.pipeline import default_audioldm_config
    P = ru.default_audioldm_config(model_name)["model"]["params"]
    cond = default_audioldm_config(model_name)["model"]["params"]["cond_stage_config"]
    for k in cond:
        cond[k]["params"]["device"] = "cpu"
    P["cond_stage_config"] = cond
    P["device"] = "cpu"
    torch.manual_seed(0)
    ld = LatentDiffusion(**P).eval()
    hot = {k: tuple(v.shape) for k, v in ld.state_dict().items()
           if k.startswith("model.diffusion_model.") or k.startswith("first_stage_model.")}
    new = weights.make_state_dict(hot, seed=0)
    new["scale_factor"] = torch.tensor(cases.SCALE_FACTOR)
    missing, unexpected = ld.load_state_dict(new, strict=False)
    assert not unexpected
    with open(os.path.join(OUT, keys_json), "w") as f:
        json.dump({k: list(v) for k, v in hot.items()}, f)
    return ld


def gen_e2e_48k(steps: int, B: int, name: str):
    """BASELINE config 3: reference generate_batch of audioldm_48k (FiLM-conditioned UNet on a [16,128,32]
    latent, 4-level VAE, 48 kHz HiFi-GAN).  The 491 536-sample waveform is stored decimated (head + every
    16th sample) to keep the fixture small."""
    ld = _ref_latent_diffusion_named("audioldm_48k", "e2e48k_statedict_keys.json")
    ld.latent_t_size = 128
    rec = {}
    orig_decode = ld.decode_first_stage

    def decode_hook(z):
        rec["latent"] = z.clone()
        return orig_decode(z)
    ld.decode_first_stage = decode_hook
    _seed_all()
    t0 = time.time()
    batch = cases.e2e_batch(B)
    batch["log_mel_spec"] = torch.zeros((B, 1024, 256))
    batch["fbank"] = batch["log_mel_spec"]
    wav = ld.generate_batch(batch, unconditional_guidance_scale=3.5, ddim_steps=steps, n_gen=1, duration=10)
    print(f"{name}: reference generate_batch(48k) B={B} steps={steps}: {time.time()-t0:.1f}s wave {wav.shape} "
          f"rms {np.sqrt((wav**2).mean()):.4f} latent std {rec['latent'].std():.3f}")
    ld.decode_first_stage = orig_decode
    btw = between_sample_rms(ld, wav, rec["latent"])
    print(f"{name}: between-sample wave rms {btw:.4f} = {btw / rms64(wav):.2f} x wave rms")
    save(name, latent=rec["latent"], wave_head=wav[..., :32768], wave_dec=wav[..., ::16],
         wave_len=np.int64(wav.shape[-1]), wave_rms=np.float64(rms64(wav)), wave_between_rms=np.float64(btw))


def gen_e2e_named(model_name: str, steps: int, B: int, name: str, keys_json: str):
    """BASELINE configs 4 / 5: reference generate_batch of `audioldm2-full-large-1150k` (three context slots,
    transformer depth 2; utils.py:118-120) or `audioldm2-speech-gigaspeech` (one 512-token AudioMAE context,
    utils.py:121-187), 16 kHz VAE + vocoder.  Waveform stored like the 48 kHz fixture (head + every 16th sample)."""
    ld = _ref_latent_diffusion_named(model_name, keys_json)
    ld.latent_t_size = 256
    rec = {}
    orig_decode = ld.decode_first_stage

    def decode_hook(z):
        rec["latent"] = z.clone()
        return orig_decode(z)
    ld.decode_first_stage = decode_hook
    _seed_all()
    t0 = time.time()
    wav = ld.generate_batch(cases.e2e_batch(B), unconditional_guidance_scale=3.5, ddim_steps=steps, n_gen=1, duration=10)
    print(f"{name}: reference generate_batch({model_name}) B={B} steps={steps}: {time.time()-t0:.1f}s wave {wav.shape} "
          f"rms {np.sqrt((wav**2).mean()):.4f} latent std {rec['latent'].std():.3f}")
    ld.decode_first_stage = orig_decode
    btw = between_sample_rms(ld, wav, rec["latent"])
    print(f"{name}: between-sample wave rms {btw:.4f} = {btw / rms64(wav):.2f} x wave rms")
    save(name, latent=rec["latent"], wave_head=wav[..., :32768], wave_dec=wav[..., ::16],
         wave_len=np.int64(wav.shape[-1]), wave_rms=np.float64(rms64(wav)), wave_between_rms=np.float64(btw))


def gen_seqgen():
    """§8(f) rank 1: the REAL Sequence2AudioMAE.generate (sequence_input.py:294-325; transformers GPT2Model underneath) for
    the full model's generator configuration (8 tokens from CLAP + T5) and the speech model's (CLAP + phonemes; 24 of
    its 512 steps), deterministic weights, padded conditioning."""
    keys = None
    for name, cfg, B, T in (("seqgen_full_8step_b2", cases.SEQGEN_FULL, 2, 20), ("seqgen_speech_24step_b2", cases.SEQGEN_SPEECH, 2, 40)):
        m = refimport.sequence_generator(cfg["steps"], cfg["keys"], cfg["dims"])
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if k != "model.wte.weight"}  # token table: unused
        sd = weights.make_state_dict(shapes, seed=0)
        assert m.load_state_dict(sd, strict=False).missing_keys == ["model.wte.weight"]
        cond = cases.seqgen_cond(cfg, B, T)
        t0 = time.time()
        out, _ = m.generate(None, cond_dict=cond)
        print(f"{name}: reference generate steps={cfg['steps']} B={B}: {time.time()-t0:.1f}s out {tuple(out.shape)} std {out.std():.3f}")
        save(name, out=out)
        with open(os.path.join(OUT, name + "_keys.json"), "w") as f:
            json.dump({k: list(v) for k, v in shapes.items()}, f)


def gen_phoneme():
    """§8(f) rank 1, second half: the REAL PhonemeEncoder (encoders/modules.py:30-110 over phoneme_encoder/encoder.py and
    attentions.py) at the speech model's configuration, deterministic weights: a padded batch and the unconditional
    (all-pad) condition."""
    m = refimport.phoneme_encoder(**cases.PHONEME)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = cases.phoneme_state_dict(shapes)
    m.load_state_dict(sd, strict=True)
    idx = cases.phoneme_input()
    t0 = time.time()
    emb, mask = m(idx)
    uemb, umask = m.get_unconditional_condition(2)
    print(f"phoneme: reference PhonemeEncoder B={idx.shape[0]}: {time.time()-t0:.2f}s emb {tuple(emb.shape)} std {emb.std():.3f} "
          f"mask sum {mask.sum(-1).tolist()}")
    save("phoneme_speech_b4", emb=emb, mask=mask, uncond_emb=uemb, uncond_mask=umask)
    with open(os.path.join(OUT, "phoneme_keys.json"), "w") as f:
        json.dump({k: list(v) for k, v in shapes.items()}, f)


def gen_t5():
    """§8(f) rank 2: the REAL FlanT5HiddenState.encode_text / get_unconditional_condition (encoders/modules.py:138-198;
    transformers T5EncoderModel underneath) at flan-t5-large's geometry with 3 layers, deterministic weights, a padded
    token batch (the tokenizer is replaced by fixed ids: oracle/refimport.py)."""
    cfg = cases.t5_test_config()
    ids, mask = cases.t5_tokens()

    def tok(prompt):
        if list(prompt) == [""]:
            return ids[2:3, :1], mask[2:3, :1]   # the empty prompt tokenises to EOS alone
        return ids, mask
    m = refimport.flan_t5_hidden_state(cfg, tok)
    shapes = {k: tuple(v.shape) for k, v in m.model.state_dict().items()}
    sd = cases.t5_state_dict(shapes)
    m.model.load_state_dict(sd, strict=True)
    hs, am = m(["a", "b", "c"])
    uhs, uam = m.get_unconditional_condition(2)
    print(f"t5: reference FlanT5HiddenState B=3 T={ids.shape[1]}: hidden {tuple(hs.shape)} std {hs.std():.3f}; uncond {tuple(uhs.shape)}")
    save("t5_large3_b3", hidden=hs, mask=am, uncond_hidden=uhs, uncond_mask=uam)
    with open(os.path.join(OUT, "t5_keys.json"), "w") as f:
        json.dump({k: list(v) for k, v in shapes.items()}, f)


def gen_clap_text():
    """§8(f) rank 2, second half: the CLAP text tower as clap/open_clip/model.py:513-529 builds it (transformers RobertaModel +
    the Linear-ReLU-Linear projection) evaluated as :656-663 / :730-747 evaluate it, roberta-base geometry with 2 layers,
    deterministic weights, padded token batch."""
    import torch.nn as nn
    import torch.nn.functional as F
    from transformers import RobertaConfig, RobertaModel
    cfg = cases.clap_text_test_config()
    branch = RobertaModel(RobertaConfig(**cfg)).eval()
    proj = nn.Sequential(nn.Linear(768, 512), nn.ReLU(), nn.Linear(512, 512)).eval()
    shapes = {"text_branch." + k: tuple(v.shape) for k, v in branch.state_dict().items() if v.is_floating_point()}
    shapes.update({"text_projection." + k: tuple(v.shape) for k, v in proj.state_dict().items()})
    sd = weights.make_state_dict(shapes, seed=0)
    branch.load_state_dict({k[len("text_branch."):]: v for k, v in sd.items() if k.startswith("text_branch.")}, strict=False)
    proj.load_state_dict({k[len("text_projection."):]: v for k, v in sd.items() if k.startswith("text_projection.")})
    ids, mask = cases.clap_text_tokens()
    x = branch(input_ids=ids, attention_mask=mask)["pooler_output"]
    emb = F.normalize(proj(x), dim=-1)
    print(f"clap_text: RobertaModel + projection B=3 T={ids.shape[1]}: emb {tuple(emb.shape)} pooled std {x.std():.3f}")
    save("clap_text_base2_b3", emb=emb, pooled=x)
    with open(os.path.join(OUT, "clap_text_keys.json"), "w") as f:
        json.dump({k: list(v) for k, v in shapes.items()}, f)


def gen_htsat():
    """§8(f) rank 4: the REAL `HTSAT_Swin_Transformer` (clap/open_clip/htsat.py) at HTSAT-base's geometry with depths (2,2,2,2),
    its two torchlibrosa extractors (absent here) replaced by the restated ones of oracle/htsat.py, + the reference's
    audio_projection head (model.py:563-567): 16 kHz waveforms -> resample -> embedding -> normalised CLAP audio embedding."""
    import torch.nn as nn
    import torch.nn.functional as F
    from oracle import htsat as oh
    hc = cases.htsat_test_config()
    ac = dict(oh.AUDIO_CFG)
    m = refimport.htsat_swin_transformer(hc, ac)
    proj = nn.Sequential(nn.Linear(8 * hc["embed_dim"], 512), nn.ReLU(), nn.Linear(512, 512)).eval()
    skip = ("relative_position_index", "attn_mask", "num_batches_tracked", "tscam_conv", "head.")
    shapes = {"audio_branch." + k: tuple(v.shape) for k, v in m.state_dict().items() if not any(t in k for t in skip)}
    shapes.update({"audio_projection." + k: tuple(v.shape) for k, v in proj.state_dict().items()})
    sd = cases.htsat_state_dict(shapes)
    missing = m.load_state_dict({k[len("audio_branch."):]: v for k, v in sd.items() if k.startswith("audio_branch.")}, strict=False)
    assert all(any(t in k for t in skip) for k in missing.missing_keys), missing.missing_keys
    proj.load_state_dict({k[len("audio_projection."):]: v for k, v in sd.items() if k.startswith("audio_projection.")})
    wav16 = cases.clap_waveform(2)
    t0 = time.time()
    wav48 = oh.resample(wav16, 16000, 48000)[:, : ac["clip_samples"]]
    out = m({"waveform": wav48}, device="cpu")
    emb = F.normalize(proj(out["embedding"]), dim=-1)
    print(f"htsat: reference HTSAT_Swin_Transformer depths {hc['depths']} B=2: {time.time()-t0:.1f}s embedding {tuple(out['embedding'].shape)} "
          f"std {out['embedding'].std():.3f}")
    save("htsat_base2222_b2", embedding=out["embedding"], emb=emb, wav48_head=wav48[:, :4096])
    with open(os.path.join(OUT, "htsat_keys.json"), "w") as f:
        json.dump({k: list(v) for k, v in shapes.items()}, f)


def gen_e2e_cond(steps: int, name: str):
    """the path prompts -> conditioners -> sampler -> waveform as ONE job of the REAL reference classes —
    `LatentDiffusion.generate_batch` (ddpm.py:1477) with `cond_stage_config` = audioldm2-full's (utils.py:354-411):
    SequenceGenAudioMAECond (real) over CLAP text (oracle/refcond.RefClapText), FlanT5HiddenState (real, 3 layers) and the
    AudioMAE stand-in, plus the outer FlanT5HiddenState.  Two different prompts, CFG 3.5, seed 42."""
    from oracle import refcond
    refcond.encoders_modules()
    import audioldm2.utils as ru
    from audioldm2.latent_diffusion.models.ddpm import LatentDiffusion
    P = ru.default_audioldm_config("audioldm2-full")["model"]["params"]
    P["cond_stage_config"] = refcond.cond_stage_config()
    P["device"] = "cpu"
    torch.manual_seed(0)
    ld = LatentDiffusion(**P).eval()
    sd = ld.state_dict()
    hot = {k: tuple(v.shape) for k, v in sd.items()
           if k.startswith("model.diffusion_model.") or k.startswith("first_stage_model.")}
    cond = {k: tuple(v.shape) for k, v in sd.items() if k.startswith("cond_stage_models.") and v.is_floating_point()
            and not k.endswith("model.wte.weight")}   # GPT-2's token table is never used (inputs_embeds)
    new = weights.make_state_dict(hot, seed=0)
    new.update(cases.cond_state_dict(cond, seed=0))
    new["scale_factor"] = torch.tensor(cases.SCALE_FACTOR)
    missing, unexpected = ld.load_state_dict(new, strict=False)
    assert not unexpected, unexpected
    with open(os.path.join(OUT, "e2econd_statedict_keys.json"), "w") as f:
        json.dump({"hot": {k: list(v) for k, v in hot.items()}, "cond": {k: list(v) for k, v in cond.items()}}, f)
    ld.latent_t_size = 256
    rec = {}
    orig_decode = ld.decode_first_stage
    seq = ld.cond_stage_models[0]
    orig_forward = seq.forward

    def decode_hook(z):
        rec["latent"] = z.clone()
        return orig_decode(z)

    def forward_hook(batch):
        ret = orig_forward(batch)
        rec["tokens"] = ret["crossattn_audiomae_generated"][0].clone()
        rec["clap"] = ret["film_clap_cond1"].clone()
        rec["t5"], rec["t5_mask"] = ret["crossattn_flan_t5"][0].clone(), ret["crossattn_flan_t5"][1].clone()
        return ret
    ld.decode_first_stage = decode_hook
    seq.forward = forward_hook
    _seed_all()
    t0 = time.time()
    wav = ld.generate_batch(cases.e2e_cond_batch(), unconditional_guidance_scale=3.5, ddim_steps=steps, n_gen=1, duration=10)
    print(f"{name}: reference generate_batch with the real conditioner stack, B={wav.shape[0]} steps={steps}: {time.time()-t0:.1f}s "
          f"wave rms {rms64(wav):.4f} tokens std {rec['tokens'].std():.3f} latent std {rec['latent'].std():.3f}")
    ld.decode_first_stage = orig_decode
    btw = between_sample_rms(ld, wav, rec["latent"])
    save(name, latent=rec["latent"], tokens=rec["tokens"], clap=rec["clap"], t5=rec["t5"], t5_mask=rec["t5_mask"],
         wave_head=wav[..., :32768], wave_dec=wav[..., ::16], wave_len=np.int64(wav.shape[-1]), wave_rms=np.float64(rms64(wav)),
         wave_between_rms=np.float64(btw))


def _seed_all():
    import random
    random.seed(cases.E2E_SEED)
    np.random.seed(cases.E2E_SEED)
    torch.manual_seed(cases.E2E_SEED)


def gen_e2e_masked(steps: int, B: int, name: str):
    """Reference LatentDiffusion.generate_batch_masked (ddpm.py:1573-1676): VAE-encode a real mel,
    time/frequency mask, DDIM with the q_sample blend (ddim.py:226-231), decode, vocode."""
    ld = _ref_latent_diffusion()
    ld.latent_t_size = 256
    rec = {}
    orig_decode, orig_sample_log = ld.decode_first_stage, ld.sample_log

    def decode_hook(z):
        rec["latent"] = z.clone()
        mel = orig_decode(z)
        rec["mel"] = mel.clone()
        return mel

    def sample_log_hook(*a, **k):
        rec["x0"] = k["x0"].clone()
        rec["mask"] = k["mask"].clone()
        return orig_sample_log(*a, **k)
    ld.decode_first_stage, ld.sample_log = decode_hook, sample_log_hook
    _seed_all()
    t0 = time.time()
    wav = ld.generate_batch_masked(cases.e2e_masked_batch(B), unconditional_guidance_scale=2.5, ddim_steps=steps,
                                   n_gen=1, duration=10, time_mask_ratio_start_and_end=(0.25, 0.75),
                                   freq_mask_ratio_start_and_end=(0.75, 1.0))
    dt = time.time() - t0
    print(f"{name}: reference generate_batch_masked B={B} steps={steps}: {dt:.1f}s wave rms {np.sqrt((wav**2).mean()):.4f}"
          f" latent std {rec['latent'].std():.3f} x0 std {rec['x0'].std():.3f}")
    ld.decode_first_stage, ld.sample_log = orig_decode, orig_sample_log
    btw = between_sample_rms(ld, wav, rec["latent"])
    save(name, x0=rec["x0"], mask=rec["mask"], latent=rec["latent"], mel=rec["mel"], wave=wav, wave_between_rms=np.float64(btw))


def gen_ancestral(T: int, B: int, name: str):
    """Reference LatentDiffusion.sample -> p_sample_loop (ddpm.py:1350-1391, 1276-1347) for the last T
    timesteps (timesteps=T), no CFG (the ancestral path has none)."""
    ld = _ref_latent_diffusion()
    ld.latent_t_size = 256
    batch = cases.e2e_batch(B)
    cond = {}
    for key, meta in ld.cond_stage_model_metadata.items():
        m = ld.cond_stage_models[meta["model_idx"]]
        cond[key] = m(batch if meta["cond_stage_key"] == "all" else batch[meta["cond_stage_key"]])
    _seed_all()
    t0 = time.time()
    z, inter = ld.sample(cond, batch_size=B, return_intermediates=True, timesteps=T, verbose=False, log_every_t=1)
    print(f"{name}: reference sample(timesteps={T}) B={B}: {time.time()-t0:.1f}s latent std {z.std():.3f}")
    save(name, latent=z, first=inter[1])


if __name__ == "__main__":
    what = sys.argv[1:] or ["all"]
    torch.set_grad_enabled(False)
    if "all" in what or "ddim" in what:
        gen_ddim()
    if "all" in what or "unet" in what:
        gen_unet()
    if "all" in what or "vae" in what:
        gen_vae()
    if "all" in what or "hifigan" in what:
        gen_hifigan()
    if "all" in what or "stft" in what:
        gen_stft()
    if "all" in what or "e2e5" in what:
        gen_e2e(5, 2, "e2e_full_5step_b2")
    if "all" in what or "masked" in what:
        gen_e2e_masked(4, 1, "e2e_masked_4step_b1")
    if "all" in what or "e2e48k" in what:
        gen_e2e_48k(2, 1, "e2e_48k_2step_b1")
    if "all" in what or "ancestral" in what:
        gen_ancestral(4, 1, "ancestral_4step_b1")
    if "all" in what or "e2espeech" in what:
        gen_e2e_named("audioldm2-speech-gigaspeech", 2, 1, "e2e_speech_2step_b1", "e2espeech_statedict_keys.json")
    if "all" in what or "e2elarge" in what:
        gen_e2e_named("audioldm2-full-large-1150k", 2, 1, "e2e_large_2step_b1", "e2elarge_statedict_keys.json")
    if "all" in what or "seqgen" in what:
        gen_seqgen()
    if "all" in what or "phoneme" in what:
        gen_phoneme()
    if "all" in what or "t5" in what:
        gen_t5()
    if "all" in what or "clap_text" in what:
        gen_clap_text()
    if "all" in what or "htsat" in what:
        gen_htsat()
    # round 3 (VERDICT r2 next #3e): BASELINE config 2's batch, and configs 3 / 4 / 5 at 20 steps, B = 2
    if "all" in what or "e2e5b8" in what:
        gen_e2e(5, 8, "e2e_full_5step_b8", decimate=True)
    if "all" in what or "e2e48k20" in what:
        gen_e2e_48k(20, 2, "e2e_48k_20step_b2")
    if "all" in what or "e2espeech20" in what:
        gen_e2e_named("audioldm2-speech-gigaspeech", 20, 2, "e2e_speech_20step_b2", "e2espeech_statedict_keys.json")
    if "all" in what or "e2elarge20" in what:
        gen_e2e_named("audioldm2-full-large-1150k", 20, 2, "e2e_large_20step_b2", "e2elarge_statedict_keys.json")
    if "all" in what or "e2econd" in what:
        gen_e2e_cond(4, "e2e_cond_4step_b2")
    # round 4 (VERDICT r3 next #1b): configs 3 / 4 / 5 at the BENCH batch (8 prompts => 16-sample CFG passes), 5 steps — the
    # instantiations the tuned tables pick at M = 16 samples are the ones bench.py's `configs` numbers come from
    if "all" in what or "e2e48k5b8" in what:
        gen_e2e_48k(5, 8, "e2e_48k_5step_b8")
    if "all" in what or "e2espeech5b8" in what:
        gen_e2e_named("audioldm2-speech-gigaspeech", 5, 8, "e2e_speech_5step_b8", "e2espeech_statedict_keys.json")
    if "all" in what or "e2elarge5b8" in what:
        gen_e2e_named("audioldm2-full-large-1150k", 5, 8, "e2e_large_5step_b8", "e2elarge_statedict_keys.json")
    if "e2e200" in what:
        gen_e2e(200, 1, "e2e_full_200step_b1")
