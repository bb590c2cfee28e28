"""GPU parity tests proper: the HIP product (through the C ABI) against
  (1) the committed fixtures produced by the REAL reference (tests/golden, oracle/make_golden.py),
  (2) the CPU oracle on fresh seeded inputs,
  (3) size-independent properties at the BASELINE batch (batch-invariance, determinism,
      CFG-batched == two passes, HIP-graph replay == eager).
Tolerances are PER PRODUCT MODE and at most 5x the measured error of that mode (tests/tolerances.py): in
the fp32-grade modes a UNet forward must be within 1e-5 max-norm of the real reference's fixture, VAE / HiFi-GAN within 4e-5, a
5-step latent within 1e-5 relative rms, the 200-step latent within 5e-6 and its mel within 1e-5 — bars the opt-in bf16x3 mode
(1e-5-level errors) does NOT meet and therefore does not share: it keeps 2e-4 / 1e-4.  Waveform after 200 DDIM steps: RMS error
< 1e-3 (the north-star bound) AND < 1e-3 of the rms distance between two unrelated samples.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cases, weights
from tolerances import latent_tol, log_err, mel_tol, tail_tol, unet_tol

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def batch_tol():
    """Agreement of the same computation at two batch sizes: the launches pick different tiles / split-K / attention
    variants, i.e. different fp32 summation orders (1e-7-level differences).  In the bf16x3 mode those move the
    round-to-nearest (hi, mid) operand splits of later layers by an ulp of `mid` here and there, so the two runs agree to
    that mode's own noise floor (per-GEMM 4e-6 rms) rather than to 1e-5."""
    from audioldm2_amd import ops
    return 5e-5 if ops.MMA_MODE == "bf16x3" else 1e-5


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


# The product modes the parity tests pin END TO END: "bf16x6" — the library default, fp32-grade, the mode
# bench.py's headline is measured in — and "bf16x3", the opt-in fast mode behind bench.py's `fast` sub-record.  Tests without the
# parameter run the default.
MODES = ["bf16x6", "bf16x3", "f16x3"]   # f16x3 (round 6): held to the fp32-grade bars, like bf16x6


@pytest.fixture(params=MODES)
def mma(request):
    from audioldm2_amd import ops
    prev = ops.set_mma(request.param)
    yield request.param
    ops.set_mma(prev)


def rel(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def report(line):
    """Parity numbers for DESIGN.md: echoed and appended to gpurun_out/parity_report.txt."""
    print(line)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_report.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt((a ** 2).mean()))


def load_det(module, seed=0):
    module.load_state_dict(weights.make_state_dict(weights.shapes_of(module), seed=seed))
    return module


def cu(x):
    if x is None:
        return None
    if isinstance(x, (list, tuple)):
        return [cu(e) for e in x]
    return x.cuda()


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,cfg,B,H,W,t5", [
    ("unet_tiny", cases.UNET_TINY, 2, 16, 8, 12),
    ("unet_large_tiny", cases.UNET_LARGE_TINY, 2, 16, 8, 12),
    ("unet_film_tiny", cases.UNET_FILM_TINY, 2, 8, 16, 12),
    ("unet_full", cases.UNET_FULL, 1, 256, 16, 32),
])
def test_unet_matches_reference_fixture(name, cfg, B, H, W, t5, mma):
    from audioldm2_amd.unet import UNetModel
    m = load_det(UNetModel(**cfg))
    x, t, ctxs, masks, y = cases.unet_inputs(cfg, B, H, W, t5)
    out = m(x.cuda(), t.cuda(), y=cu(y), context_list=cu(ctxs), context_attn_mask_list=cu(masks))
    e = rel(out, gold(name)["out"])
    report(f"{name} [{mma}]: max-norm rel err vs reference fixture {e:.2e}")
    assert log_err(e, unet_tol(mma), name) < unet_tol(mma)


@pytest.mark.parametrize("name,cfg,B,H,W,t5", [("unet_tiny", cases.UNET_TINY, 2, 16, 8, 12),
                                                ("unet_film_tiny", cases.UNET_FILM_TINY, 2, 8, 16, 12)])
def test_unet_matches_reference_fixture_on_fp32_mfma_path(name, cfg, B, H, W, t5):
    """The default matrix-core path is the bf16-split one (every other model test); the plain fp32-MFMA kernels
    (ALDM_MMA=f32) stay a supported configuration and meet the same fixtures."""
    from audioldm2_amd import ops
    from audioldm2_amd.unet import UNetModel
    prev = ops.set_mma("f32")
    try:
        m = load_det(UNetModel(**cfg))
        x, t, ctxs, masks, y = cases.unet_inputs(cfg, B, H, W, t5)
        out = m(x.cuda(), t.cuda(), y=cu(y), context_list=cu(ctxs), context_attn_mask_list=cu(masks))
    finally:
        ops.set_mma(prev)
    e = rel(out, gold(name)["out"])
    report(f"{name} (fp32 MFMA path): max-norm rel err vs reference fixture {e:.2e}")
    assert log_err(e, unet_tol("f32"), name + " f32") < unet_tol("f32")


def test_unet_full_vs_oracle_batch8_properties():
    """BASELINE batch (8 prompts, CFG => 16 rows): agrees with the CPU oracle on two rows, is
    batch-invariant (row b of the batch-16 pass == the same row run alone) and deterministic."""
    from audioldm2_amd.unet import UNetModel
    from oracle.unet import unet_forward
    cfg = cases.UNET_FULL
    m = load_det(UNetModel(**cfg))
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    x, t, ctxs, masks, _ = cases.unet_inputs(cfg, 16, 256, 16, 32, seed=3)
    out = m(x.cuda(), t.cuda(), context_list=cu(ctxs), context_attn_mask_list=cu(masks))
    out2 = m(x.cuda(), t.cuda(), context_list=cu(ctxs), context_attn_mask_list=cu(masks))
    assert torch.equal(out, out2), "non-deterministic UNet forward"
    for b in (0, 11):
        sl = slice(b, b + 1)
        ref = unet_forward(sd, cfg, x[sl], t[sl], [c[sl] for c in ctxs], [mm[sl] for mm in masks])
        assert log_err(rel(out[sl], ref), unet_tol(), "unet_full b16 vs oracle") < unet_tol()
        alone = m(x[sl].cuda(), t[sl].cuda(), context_list=cu([c[sl] for c in ctxs]),
                  context_attn_mask_list=cu([mm[sl] for mm in masks]))
        assert rel(alone, out[sl]) < batch_tol()  # same kernels; only tile/grid shapes differ


def test_unet_full_batch16_vs_pytorch_rocm_eager():
    """BASELINE config 2 size (8 prompts x CFG = 16 samples, 256x16 latent): the HIP UNet against the
    oracle's functional restatement executed by stock PyTorch-ROCm ON THE SAME GPU (ATen / rocBLAS /
    MIOpen, fp32) — every one of the 16 rows, tolerance 2e-4 max-norm relative — and both timed (the
    PyTorch-ROCm eager number is the 'reference stack on this hardware' context for DESIGN.md)."""
    from audioldm2_amd.unet import UNetModel
    from oracle.unet import unet_forward
    cfg = cases.UNET_FULL
    m = load_det(UNetModel(**cfg))
    sd = {k: v.detach().cuda() for k, v in m.state_dict().items()}
    x, t, ctxs, masks, _ = cases.unet_inputs(cfg, 16, 256, 16, 32, seed=4)
    xg, tg, cg, mg = x.cuda(), t.cuda(), cu(ctxs), cu(masks)

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        return out, e0.elapsed_time(e1) / n
    with torch.no_grad():  # HIP path first: it is timed in the state a sampling job sees (nothing else resident)
        out, t_hip = timed(lambda: m(xg, tg, context_list=cg, context_attn_mask_list=mg), 3)
        ref, t_torch = timed(lambda: unet_forward(sd, cfg, xg, tg, cg, mg), 3)
    e = rel(out, ref)
    report(f"unet_full B=16 on MI355X: HIP path {t_hip:.1f} ms/pass (eager launches) vs PyTorch-ROCm eager fp32 "
           f"{t_torch:.1f} ms/pass; max-norm rel err {e:.2e}")
    assert log_err(e, unet_tol(), "unet_full b16 vs rocm eager") < unet_tol()


@pytest.mark.parametrize("name,dd,shapes", [("vae16k", cases.DDCONFIG_16K, [(2, 8, 32, 16), (1, 8, 256, 16)]),
                                            ("vae48k", cases.DDCONFIG_48K, [(1, 16, 16, 32)])])
def test_vae_matches_reference_fixture(name, dd, shapes):
    from audioldm2_amd.vae import AutoencoderKL
    ae = load_det(AutoencoderKL(ddconfig=dd, embed_dim=dd["z_channels"], image_key="fbank"))
    g = gold(name)
    for i, shp in enumerate(shapes):
        mel = ae.decode(cases.latent_input(*shp, seed=i).cuda())
        e = rel(mel, g[f"mel{i}"])
        report(f"{name} decode {shp}: max-norm rel err vs reference fixture {e:.2e}")
        assert log_err(e, tail_tol(), f"{name} decode") < tail_tol()
    f = 2 ** (len(dd["ch_mult"]) - 1)
    x = cases.mel_input(1, dd["mel_bins"], 16 * f, seed=5).permute(0, 2, 1)[:, None].contiguous()
    post = ae.encode(x.cuda())
    assert log_err(rel(post.parameters, g["moments"]), tail_tol(), f"{name} encode") < tail_tol()


@pytest.mark.parametrize("name,hc,Ts", [("hifigan16k", cases.HIFIGAN_16K, [48, 1024]),
                                        ("hifigan48k", cases.HIFIGAN_48K, [24])])
def test_hifigan_matches_reference_fixture(name, hc, Ts):
    from audioldm2_amd.hifigan import Generator
    gen = load_det(Generator(dict(hc)))
    g = gold(name)
    for i, T in enumerate(Ts):
        w = gen(cases.mel_input(1, hc["num_mels"], T, seed=i).cuda())
        assert tuple(w.shape) == tuple(g[f"wave{i}"].shape)
        e = rel(w, g[f"wave{i}"])
        report(f"{name} T={T}: max-norm rel err vs reference fixture {e:.2e}")
        assert log_err(e, tail_tol(), f"{name} T={T}") < tail_tol()


def test_stft_mel_matches_reference_fixture_and_oracle():
    from audioldm2_amd.stft import TacotronSTFT
    from oracle import stft as ostft
    st = TacotronSTFT(1024, 160, 1024, 64, 16000, 0, 8000)
    g = gold("stft16k")
    assert np.array_equal(st.mel_basis.numpy(), g["mel_basis"])
    x = cases.wave_input(2, 16000, seed=0)
    mel, mag, phase, energy = st.mel_spectrogram(x)
    assert rel(mag, g["mag"]) < 5e-5 and rel(energy, g["energy"]) < 5e-5
    gm = torch.from_numpy(g["mel"])
    assert rel(mel.exp(), gm.exp()) < 5e-5  # linear mel energies
    loud = gm > gm.max() - 9.0  # log domain only where the bin is not ~1e-4 of the peak (fp32 leakage noise)
    assert float((mel - gm)[loud].abs().max()) < 2e-3
    # full 10.24 s clip (BASELINE shape [B, 163840]) against the oracle, incl. phase where |X| is not tiny
    x = cases.wave_input(2, 163840, seed=1)
    mel, mag, phase, energy = st.mel_spectrogram(x)
    omel, omag, ophase, oen = ostft.mel_spectrogram(x)
    assert tuple(mel.shape) == (2, 64, 1025) and rel(mag, omag) < 5e-5
    big = omag > 1e-2 * omag.max()
    dphi = torch.remainder(phase - ophase + np.pi, 2 * np.pi) - np.pi
    assert float(dphi[big].abs().max()) < 1e-3
    st48 = TacotronSTFT(2048, 480, 2048, 256, 48000, 20, 24000)
    x48 = cases.wave_input(1, 48000, seed=2)
    mel48 = st48.mel_spectrogram(x48)[0]
    omel48 = ostft.mel_spectrogram(x48, 2048, 480, 2048, 256, 48000, 20, 24000)[0]
    assert rel(mel48.exp(), omel48.exp()) < 5e-5


# ---------------------------------------------------------------------------------------------
def _build_ld(t5_len=32):
    from audioldm2_amd.pipeline import build_model
    ld = build_model(model_name="audioldm2-full")
    with open(os.path.join(GOLD, "e2e_statedict_keys.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = weights.make_state_dict(shapes, seed=0)
    sd["scale_factor"] = torch.tensor(cases.SCALE_FACTOR)
    ld.load_state_dict(sd, strict=False)
    return ld.cuda()


@pytest.fixture(scope="module")
def ld():
    return _build_ld()


@pytest.fixture
def ld_mode(ld, mma):
    """The module's LatentDiffusion with the product mode switched: the captured step graph and the cached cross-attention
    K/V projections belong to a mode, the packed weights keep one split image per mode."""
    ld.model.diffusion_model.drop_step_caches()
    yield ld, mma
    ld.model.diffusion_model.drop_step_caches()


# Waveform tolerance (VERDICT r2 "weak" #1): north_star's "within 1e-3 rms" is read against what separates two DIFFERENT
# samples — every fixture stores `wave_between_rms`, the rms difference between the waveforms of two unrelated samples on
# these (random-init) weights — so a pass means "1000x closer to the reference than another sample would be", not merely
# "smaller than a constant the vocoder's biases already satisfy".  All parity here is on random-init weights, in the library
# default mode (bf16x6, fp32-grade) unless a test is parametrised over MODES or $ALDM_MMA says otherwise.
WAVE_REL_TO_BETWEEN = 1e-3


def _assert_wave(err_rms, g, what=""):
    between = float(g["wave_between_rms"])
    assert between > 1e-2, "fixture waveforms must depend on the sample"
    assert err_rms < 1e-3, (what, err_rms)                                   # north_star, absolute
    assert err_rms < WAVE_REL_TO_BETWEEN * between, (what, err_rms, between)  # ... and relative to the between-sample distance


def _report(tag, out, g):
    errs = {}
    for k in ("latent", "mel", "wave"):
        a = np.asarray(out[k].detach().cpu() if torch.is_tensor(out[k]) else out[k], dtype=np.float64)
        b = g[k].astype(np.float64)
        errs[k] = (rms(a - b), rms(b))
    report(f"{tag}: " + "  ".join(f"{k}: rms_err {e:.3e} / rms_ref {r:.3e} (rel {e/r:.2e})" for k, (e, r) in errs.items()))
    return errs


def _generate(ld, B, steps):
    from audioldm2_amd.pipeline import seed_everything
    rec = {}
    orig = ld.decode_first_stage_cl

    def hook(z):
        rec["latent"] = z.clone()
        mel = orig(z)
        rec["mel"] = mel.view(mel.shape[0], 1, mel.shape[1], mel.shape[2]).clone()
        return mel
    ld.decode_first_stage_cl = hook
    try:
        seed_everything(cases.E2E_SEED)
        ld.latent_t_size = 256
        # every fixture is the FIRST generate_batch of a fresh reference object; the module-scoped `ld` here has run jobs before, and
        # from its second call on the reference draws one more torch.rand(1) per job (pipeline._cfg_dropout_draw; the two-jobs-in-one-
        # process behaviour is pinned by test_parity_on_checkpoint_script_hip_stage_... against the real reference)
        ld.conditional_dry_run_finished = False
        rec["wave"] = ld.generate_batch(cases.e2e_batch(B), unconditional_guidance_scale=3.5,
                                        ddim_steps=steps, n_gen=1, duration=10)
    finally:
        ld.decode_first_stage_cl = orig
    return rec


def test_e2e_5step_matches_reference_generate_batch(ld_mode):
    """Whole path vs the real LatentDiffusion.generate_batch fixture (B=2, 5 steps, CFG 3.5, seed 42):
    RNG contract, CFG batching with padded/masked contexts, DDIM update, VAE decode, vocoder.  Both product modes."""
    ld, mode = ld_mode
    g = gold("e2e_full_5step_b2")
    out = _generate(ld, 2, 5)
    assert out["wave"].dtype == np.float32 and out["wave"].shape == (2, 1, 163872)
    errs = _report(f"e2e 5 steps B=2 [{mode}]", out, g)
    assert log_err(errs["latent"][0] / errs["latent"][1], latent_tol(5, mode), "latent 5 steps B=2") < latent_tol(5, mode)
    assert log_err(errs["mel"][0] / errs["mel"][1], mel_tol(5, mode), "mel 5 steps B=2") < mel_tol(5, mode)
    _assert_wave(errs["wave"][0], g, "e2e 5 steps")


def test_e2e_5step_batch8_matches_reference_generate_batch(ld_mode):
    """BASELINE config 2's batch (8 prompts) end to end against the real reference's generate_batch (5 steps, CFG 3.5, seed
    42): the global-batch noise draws, the 16-sample CFG pass, per-sample masks (half the batch masks its last 8 T5 keys).
    Both product modes: this is the geometry (and therefore the tuned-table instantiations) bench.py's headline runs."""
    ld, mode = ld_mode
    g = gold("e2e_full_5step_b8")
    out = _generate(ld, 8, 5)
    assert out["wave"].shape == (8, 1, int(g["wave_len"]))
    el = rms(out["latent"].double().cpu().numpy() - g["latent"]) / rms(g["latent"])
    eh = rms(out["wave"][..., :32768].astype(np.float64) - g["wave_head"])
    ed = rms(out["wave"][..., ::16].astype(np.float64) - g["wave_dec"])
    report(f"e2e 5 steps B=8 [{mode}]: latent rel rms {el:.2e}  wave(head) rms_err {eh:.3e}  wave(1/16) rms_err {ed:.3e} / between-sample "
           f"{float(g['wave_between_rms']):.3e}")
    assert log_err(el, latent_tol(5, mode), "latent 5 steps B=8") < latent_tol(5, mode)
    _assert_wave(max(eh, ed), g, "e2e 5 steps B=8")


def test_cached_step_graph_is_refreshed_with_new_conditioning(ld):
    """The captured DDIM step graph, its static buffers and the cross-attention K/V caches are reused by
    the next job of the same geometry: a job with DIFFERENT conditioning and seed must not leak into
    the following seed-42 job, which has to reproduce the reference fixture through the cache-hit path."""
    from audioldm2_amd.pipeline import seed_everything
    g = gold("e2e_full_5step_b2")
    unet = ld.model.diffusion_model
    conds = list(ld.cond_stage_models)
    old = [c.seed for c in conds]
    try:
        for c in conds:
            c.seed = c.seed + 17  # other context values, same shapes
        seed_everything(123)
        ld.latent_t_size = 256
        other = ld.generate_batch(cases.e2e_batch(2), unconditional_guidance_scale=3.5, ddim_steps=5, n_gen=1,
                                  duration=10)
    finally:
        for c, s0 in zip(conds, old):
            c.seed = s0
    assert len(unet._graph_cache) == 1
    ent = next(iter(unet._graph_cache.values()))
    out = _generate(ld, 2, 5)
    assert next(iter(unet._graph_cache.values())) is ent, "second job should hit the cached graph"
    assert rms(other.astype(np.float64) - g["wave"]) > 1e-4  # the first job really was a different job
    errs = _report("e2e 5 steps B=2 via cached graph", out, g)
    assert errs["latent"][0] / errs["latent"][1] < latent_tol(5)
    _assert_wave(errs["wave"][0], g, "cached graph")


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "e2e_full_200step_b1.npz")), reason="200-step fixture absent")
def test_e2e_200step_waveform_within_north_star_tolerance(ld_mode):
    """BASELINE config 1 (1 prompt, 10 s, 200 DDIM steps, CFG 3.5, seed 42) against the reference's
    CPU run: after 400 dependent UNet evaluations the latent within 5e-6 and the mel within 1e-5 relative rms in the fp32-grade
    mode (1e-4 for bf16x3, which measures 2.7e-6 / 9.9e-6), waveform rms error < 1e-3 (north_star) AND < 1e-3 of the distance
    between two unrelated samples' waveforms."""
    ld, mode = ld_mode
    g = gold("e2e_full_200step_b1")
    out = _generate(ld, 1, 200)
    errs = _report(f"e2e 200 steps B=1 [{mode}]", out, g)
    assert log_err(errs["latent"][0] / errs["latent"][1], latent_tol(200, mode), "latent 200 steps") < latent_tol(200, mode)
    assert log_err(errs["mel"][0] / errs["mel"][1], mel_tol(200, mode), "mel 200 steps") < mel_tol(200, mode)
    _assert_wave(errs["wave"][0], g, "e2e 200 steps")


def test_generate_batch_masked_matches_reference(ld):
    """Inpainting / super-resolution path (SURVEY §8 a17/a18 consumers) vs the REAL reference's
    LatentDiffusion.generate_batch_masked fixture (B=1, 4 DDIM steps, CFG 2.5, seed 42): HIP VAE encoder,
    host posterior draw, mask, q_sample blend each step (RNG order), decode, vocoder."""
    from audioldm2_amd.pipeline import seed_everything
    g = gold("e2e_masked_4step_b1")
    rec = {}
    orig = ld.decode_first_stage_cl

    def hook(z):
        rec["latent"] = z.clone()
        return orig(z)
    ld.decode_first_stage_cl = hook
    try:
        seed_everything(cases.E2E_SEED)
        ld.latent_t_size = 256
        ld.conditional_dry_run_finished = False   # the fixture is a fresh reference object's first call (see _generate)
        wave = ld.generate_batch_masked(cases.e2e_masked_batch(1), unconditional_guidance_scale=2.5, ddim_steps=4,
                                        n_gen=1, duration=10)
    finally:
        ld.decode_first_stage_cl = orig
    el = rms(rec["latent"].double().cpu().numpy() - g["latent"]) / rms(g["latent"])
    ew = rms(wave.astype(np.float64) - g["wave"])
    report(f"masked 4 steps B=1: latent rel rms {el:.2e}  wave rms_err {ew:.3e} / rms_ref {rms(g['wave']):.3e}")
    assert wave.shape == (1, 1, 163872)
    assert log_err(el, latent_tol(4), "masked latent") < latent_tol(4)   # default (fp32-grade) mode; measured 1.3e-6
    _assert_wave(ew, g, "masked")


def test_ancestral_sample_matches_reference(ld):
    """LatentDiffusion.sample (ancestral DDPM, ddpm.py:1350-1391) vs the REAL reference fixture
    (timesteps=4, seed 42)."""
    from audioldm2_amd.pipeline import seed_everything
    g = gold("ancestral_4step_b1")
    batch = cases.e2e_batch(1)
    cond = ld.get_learned_conditioning_dict(batch)
    seed_everything(cases.E2E_SEED)
    ld.latent_t_size = 256
    z, inter = ld.sample(cond, batch_size=1, return_intermediates=True, timesteps=4, verbose=False, log_every_t=1)
    e1 = rms(inter[1].double().cpu().numpy() - g["first"]) / rms(g["first"])
    e2 = rms(z.double().cpu().numpy() - g["latent"]) / rms(g["latent"])
    report(f"ancestral 4 steps B=1: first-step rel rms {e1:.2e}  final rel rms {e2:.2e}")
    assert e1 < 2e-6 and e2 < 2e-6   # measured 2.4e-8 / 6.1e-8: one UNet pass per step feeds an exact update


def test_super_resolution_and_inpainting_entry_point(ld):
    """pipeline.super_resolution_and_inpainting (pipeline.py:213-267) end to end on a synthetic 16 kHz
    waveform: GPU STFT/mel front-end -> VAE encode -> masked DDIM -> decode -> vocoder; deterministic."""
    from audioldm2_amd.pipeline import super_resolution_and_inpainting
    wav = cases.wave_input(1, 100000, seed=3)[0].numpy()
    outs = [super_resolution_and_inpainting(ld, "a dog barking", original_audio_file_path=wav, seed=7, ddim_steps=4,
                                            duration=10, batchsize=1, guidance_scale=2.5,
                                            n_candidate_gen_per_text=1) for _ in range(2)]
    assert outs[0].shape == (1, 1, 163872) and np.isfinite(outs[0]).all()
    assert np.array_equal(outs[0], outs[1])


def test_cfg_batched_equals_two_passes_and_graph_equals_eager(ld):
    """apply_model_cfg (one 2B pass, padded + masked contexts) == two apply_model passes; the HIP
    graph replay path == the eager path (same kernels, same order => bitwise)."""
    B = 2
    batch = cases.e2e_batch(B)
    cond = ld.get_learned_conditioning_dict(batch)
    uncond = {k: ld.cond_stage_models[m["model_idx"]].get_unconditional_condition(B)
              for k, m in ld.cond_stage_model_metadata.items()}
    x = cases.latent_input(B, 8, 256, 16, seed=9).cuda()
    t = torch.tensor([501, 501])
    eps2 = ld.apply_model_cfg(x, t.float().repeat(2).cuda(), cond, uncond)
    e_u = ld.apply_model(x, t.cuda(), uncond)
    e_c = ld.apply_model(x, t.cuda(), cond)
    assert rel(eps2[0], e_u) < batch_tol() and rel(eps2[1], e_c) < batch_tol()
    os.environ["ALDM_NO_GRAPH"] = "1"
    try:
        eager = _generate(ld, 2, 4)
    finally:
        os.environ["ALDM_NO_GRAPH"] = "0"
    graph = _generate(ld, 2, 4)
    assert torch.equal(eager["latent"], graph["latent"])


def test_shared_cfg_prefix_equals_the_full_two_half_pass(ld):
    """Round 5: apply_model_cfg hands the UNet x ONCE and the UNet runs everything in front of the first transformer that receives
    a context — identical for the unconditional and the conditional half — on B samples instead of 2B (UNetModel.forward,
    cfg_shared).  Same values as the pass over the repeated batch (ALDM_CFG_SHARE=0) up to the summation orders the two batch
    sizes' tiles pick; per-sample timesteps and the padded / masked contexts included."""
    B = 3
    batch = cases.e2e_batch(B)
    cond = ld.get_learned_conditioning_dict(batch)
    uncond = {k: ld.cond_stage_models[m["model_idx"]].get_unconditional_condition(B)
              for k, m in ld.cond_stage_model_metadata.items()}
    assert ld.model.diffusion_model._shared_prefix_end([torch.zeros(1)] * 2) == (4, 2)
    x = cases.latent_input(B, 8, 256, 16, seed=21).cuda()
    t2 = torch.tensor([801.0, 401.0, 1.0]).repeat(2).cuda()
    shared = ld.apply_model_cfg(x, t2, cond, uncond)
    os.environ["ALDM_CFG_SHARE"] = "0"
    try:
        full = ld.apply_model_cfg(x, t2, cond, uncond)
    finally:
        os.environ.pop("ALDM_CFG_SHARE")
    assert shared.shape == full.shape == (2, B, 8, 256, 16)
    assert rel(shared, full) < batch_tol()
    assert rel(shared[0], shared[1]) > 1e-3   # the halves do differ (their contexts do)


def test_callback_draws_are_sequenced_with_the_step_noise_like_the_reference(ld):
    """in the reference any callback may draw from torch's default generator between two steps (ddim.py:246-249 call it
    right after p_sample_ddim's `torch.randn`, ddim.py:351).  A run that is GIVEN a callback therefore keeps the draws on the
    launching thread, in the reference's order: x_T, then per step (step noise, whatever the callback draws) — checked through the
    generator: after the run its state equals the state after exactly that sequence of CPU draws, and the latent equals the run
    whose callback draws nothing of its own but is fed the same interleaved stream.  A callback that declares
    `uses_rng = False` keeps the threaded feed (it must then really not draw)."""
    from audioldm2_amd.ddim import DDIMSampler
    B, S, shape = 2, 4, (8, 256, 16)
    batch = cases.e2e_batch(B)
    cond = ld.get_learned_conditioning_dict(batch)
    uncond = {k: ld.cond_stage_models[m["model_idx"]].get_unconditional_condition(B)
              for k, m in ld.cond_stage_model_metadata.items()}
    drawn = []

    def cb(i):                      # an ordinary reference-style callback: draws without declaring anything
        drawn.append(torch.randn(3))

    def run(callback):
        torch.manual_seed(77)
        z, _ = DDIMSampler(ld).sample(S, B, shape, cond, eta=1.0, unconditional_guidance_scale=3.5,
                                      unconditional_conditioning=uncond, verbose=False, callback=callback)
        return z.clone(), torch.get_rng_state()
    z_cb, st_cb = run(cb)
    torch.manual_seed(77)            # the reference's order of draws, on the CPU
    torch.randn((B,) + shape)
    want = []
    for _ in range(S):
        torch.randn((B,) + shape)
        want.append(torch.randn(3))
    assert torch.equal(torch.get_rng_state(), st_cb), "generator not where the reference's draw order leaves it"
    assert all(torch.equal(a, b) for a, b in zip(drawn, want)), "the callback saw other numbers than in the reference's order"
    z_cb2, _ = run(cb)
    assert torch.equal(z_cb, z_cb2)
    quiet = lambda i: None
    quiet.uses_rng = False           # declared: the threaded feed stays, and without a drawing callback the stream is x_T + S draws
    z_q, st_q = run(quiet)
    torch.manual_seed(77)
    for _ in range(S + 1):
        torch.randn((B,) + shape)
    assert torch.equal(torch.get_rng_state(), st_q)
    assert not torch.equal(z_q, z_cb)   # the callback's draws shift every later step's noise, as they do in the reference


def test_pipeline_batch8_runs_and_is_batch_consistent(ld):
    """BASELINE config 2 shape (batch 8): finite output of the right shape; prompt 0 of the batch-8 run
    equals the batch-1 run on the same noise (samples are independent: no cross-sample coupling)."""
    out8 = _generate(ld, 8, 4)
    assert out8["wave"].shape == (8, 1, 163872) and np.isfinite(out8["wave"]).all()
    # batch-1 run draws a different global noise tensor, so compare through x_T injection instead
    from audioldm2_amd.ddim import DDIMSampler
    batch = cases.e2e_batch(8)
    cond = ld.get_learned_conditioning_dict(batch)
    uncond = {k: ld.cond_stage_models[m["model_idx"]].get_unconditional_condition(8)
              for k, m in ld.cond_stage_model_metadata.items()}
    torch.manual_seed(5)
    s8, _ = DDIMSampler(ld).sample(4, 8, (8, 256, 16), cond, eta=0.0, unconditional_guidance_scale=3.5,
                                   unconditional_conditioning=uncond, verbose=False)
    cond1 = {k: [v[0][:1].contiguous(), v[1][:1].contiguous()] for k, v in cond.items()}
    unc1 = {k: [v[0][:1].contiguous(), v[1][:1].contiguous()] for k, v in uncond.items()}
    torch.manual_seed(5)
    xT = torch.randn(8, 8, 256, 16)[:1]
    s1, _ = DDIMSampler(ld).sample(4, 1, (8, 256, 16), cond1, eta=0.0, unconditional_guidance_scale=3.5,
                                   unconditional_conditioning=unc1, verbose=False, x_T=xT)
    assert rel(s1, s8[:1]) < 1e-4


@pytest.mark.parametrize("model_name,wave_len", [("audioldm_48k", 491536),
                                                   ("audioldm2-speech-gigaspeech", 163872),
                                                   ("audioldm2-full-large-1150k", 163872)])
def test_other_baseline_configs_run_end_to_end(model_name, wave_len):
    """BASELINE configs 3-5 (48 kHz FiLM-conditioned model with the 4-level VAE and the 48 k vocoder;
    speech model with 512 AudioMAE tokens; large model with transformer depth 2 and a context-free
    third transformer): text_to_audio-shaped job with 2 DDIM steps, and one 2B CFG pass must equal two
    B passes of the same HIP UNet."""
    from audioldm2_amd.pipeline import build_model, make_batch_for_text_to_audio, seed_everything
    torch.manual_seed(3)
    m = build_model(model_name=model_name).cuda()
    if torch.is_tensor(m.scale_factor):
        m.scale_factor.fill_(0.75)
    B = 2
    batch = make_batch_for_text_to_audio("a test prompt", batchsize=B)
    seed_everything(11)
    m.latent_t_size = 128 if "48k" in model_name else 256
    wav = m.generate_batch(batch, unconditional_guidance_scale=3.5, ddim_steps=2, n_gen=1, duration=10)
    assert wav.shape == (B, 1, wave_len) and wav.dtype == np.float32 and np.isfinite(wav).all()
    cond = m.get_learned_conditioning_dict(batch)
    uncond = {k: m.cond_stage_models[v["model_idx"]].get_unconditional_condition(B)
              for k, v in m.cond_stage_model_metadata.items()}
    x = torch.randn(B, m.channels, m.latent_t_size, m.latent_f_size, generator=torch.Generator().manual_seed(1)).cuda()
    t = torch.tensor([301.0, 301.0]).cuda()
    eps2 = m.apply_model_cfg(x, t.repeat(2), cond, uncond)
    e_u, e_c = m.apply_model(x, t, uncond), m.apply_model(x, t, cond)
    assert rel(eps2[0], e_u) < batch_tol() and rel(eps2[1], e_c) < batch_tol()
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("fixture,B,steps,mode", [("e2e_48k_2step_b1", 1, 2, None), ("e2e_48k_20step_b2", 2, 20, None),
                                                   ("e2e_48k_5step_b8", 8, 5, "bf16x6"), ("e2e_48k_5step_b8", 8, 5, "bf16x3"),
                                                   ("e2e_48k_5step_b8", 8, 5, "f16x3")])
def test_e2e_48k_matches_reference_generate_batch(fixture, B, steps, mode):
    """BASELINE config 3 (audioldm_48k) end to end against the REAL reference's generate_batch fixtures
    (B=1, 2 DDIM steps; B=2, 20 steps; and — VERDICT r3 next #1b — B=8, 5 steps = the bench batch, in both product modes, so the
    instantiations the tuned tables pick for 16-sample passes are the ones compared; CFG 3.5, seed 42): FiLM-conditioned UNet,
    4-level VAE decoder, 48 kHz HiFi-GAN."""
    from audioldm2_amd import ops
    from audioldm2_amd.pipeline import build_model, seed_everything
    if not os.path.exists(os.path.join(GOLD, fixture + ".npz")):
        pytest.skip(f"{fixture} absent")
    prev = ops.set_mma(mode) if mode else None
    try:
        _e2e_48k(fixture, B, steps, mode or ops.MMA_MODE)
    finally:
        if prev:
            ops.set_mma(prev)


def _e2e_48k(fixture, B, steps, mode):
    from audioldm2_amd.pipeline import build_model, seed_everything
    g = gold(fixture)
    m = build_model(model_name="audioldm_48k")
    with open(os.path.join(GOLD, "e2e48k_statedict_keys.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    sd = weights.make_state_dict(shapes, seed=0)
    sd["scale_factor"] = torch.tensor(cases.SCALE_FACTOR)
    m.load_state_dict(sd, strict=False)
    m = m.cuda()
    rec = {}
    orig = m.decode_first_stage_cl

    def hook(z):
        rec["latent"] = z.clone()
        return orig(z)
    m.decode_first_stage_cl = hook
    seed_everything(cases.E2E_SEED)
    m.latent_t_size = 128
    wave = m.generate_batch(cases.e2e_batch_48k(B), unconditional_guidance_scale=3.5, ddim_steps=steps, n_gen=1, duration=10)
    assert wave.shape == (B, 1, int(g["wave_len"]))
    el = rms(rec["latent"].double().cpu().numpy() - g["latent"]) / rms(g["latent"])
    eh = rms(wave[..., :32768].astype(np.float64) - g["wave_head"])
    ed = rms(wave[..., ::16].astype(np.float64) - g["wave_dec"])
    report(f"48k e2e {steps} steps B={B} [{mode}]: latent rel rms {el:.2e}  wave(head) rms_err {eh:.3e}  wave(1/16) rms_err {ed:.3e} "
           f"/ rms_ref {float(g['wave_rms']):.3e} / between-sample {float(g['wave_between_rms']):.3e}")
    assert log_err(el, latent_tol(steps, mode), f"48k latent {steps} steps") < latent_tol(min(steps, 5), mode)
    _assert_wave(max(eh, ed), g, fixture)
    del m
    torch.cuda.empty_cache()


def _shard_worker(rank, world, port, out_dir, gB, steps):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), ALDM_DIST_BACKEND="gloo")
    from audioldm2_amd import dist as adist
    from audioldm2_amd.pipeline import build_model, seed_everything
    r, w, _ = adist.init_distributed()
    torch.manual_seed(1000 + rank)  # different random init per rank: the broadcast must make them equal
    m = build_model(model_name="audioldm2-full").cuda()
    m.scale_factor.fill_(0.75 if rank == 0 else 0.5)
    sent = adist.broadcast_module(m, src=0)
    assert sent > 1e9 and float(m.scale_factor) == 0.75
    seed_everything(cases.E2E_SEED)
    m.latent_t_size = 256
    wav = m.generate_batch(cases.e2e_batch(gB), unconditional_guidance_scale=3.5, ddim_steps=steps, n_gen=1,
                           duration=10, shard=(rank, world))
    np.save(os.path.join(out_dir, f"wave{rank}.npy"), wav)
    if rank == 0:
        torch.save({k: v.cpu() for k, v in m.state_dict().items()}, os.path.join(out_dir, "sd.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_sharded_run_equals_single_process(tmp_path):
    """SURVEY §8e on real kernels: two ranks (sharing this one GPU, gloo instead of RCCL) broadcast the
    weights, draw the GLOBAL noise and sample contiguous halves of a 4-prompt batch; concatenated, the
    result must equal the single-process run of the same batch (prompts are independent, RNG contract R)."""
    import socket

    import torch.multiprocessing as mp
    from audioldm2_amd.pipeline import build_model, seed_everything
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    gB, steps = 4, 4  # the reference (and so this port) needs 1000 % steps == 0
    mp.spawn(_shard_worker, args=(2, port, str(tmp_path), gB, steps), nprocs=2, join=True)
    sharded = np.concatenate([np.load(os.path.join(tmp_path, f"wave{r}.npy")) for r in range(2)], axis=0)
    m = build_model(model_name="audioldm2-full")
    m.load_state_dict(torch.load(os.path.join(tmp_path, "sd.pt")), strict=False)
    m = m.cuda()
    seed_everything(cases.E2E_SEED)
    m.latent_t_size = 256
    single = m.generate_batch(cases.e2e_batch(gB), unconditional_guidance_scale=3.5, ddim_steps=steps, n_gen=1,
                              duration=10)
    assert sharded.shape == single.shape == (gB, 1, 163872)
    e = rms(sharded.astype(np.float64) - single) / rms(single)
    report(f"2-rank sharded vs single process, B={gB}, {steps} steps: wave rel rms {e:.2e}")
    assert e < batch_tol()  # same kernels; only tile/grid choices differ with the per-rank batch


def _small_clap(prob):
    """The CLAP re-ranker at test geometry (2 RoBERTa layers, HTSAT depths (2,2,2,2)), deterministic name-keyed weights, the
    stub tokenizer of oracle/cases.py."""
    from audioldm2_amd.clap import CLAPAudioEmbeddingClassifierFreev2
    with open(os.path.join(GOLD, "htsat_keys.json")) as f:
        asd = cases.htsat_state_dict({k: tuple(v) for k, v in json.load(f).items()})
    with open(os.path.join(GOLD, "clap_text_keys.json")) as f:
        tsd = weights.make_state_dict({k: tuple(v) for k, v in json.load(f).items()}, seed=0)
    clap = CLAPAudioEmbeddingClassifierFreev2(embed_mode="audio", unconditional_prob=prob, sampling_rate=16000,
                                              config=cases.clap_text_test_config(), audio_config=cases.htsat_test_config())
    clap.model.load_state_dict({**asd, **tsd}, strict=False)
    clap.tokenize = cases.StubRobertaTokenizer()
    return clap


def _cand_batch(gB):
    b = cases.e2e_batch(gB)
    b["text"] = [f"prompt number {i} of the candidate test" for i in range(gB)]
    return b


def _shard_cand_worker(rank, world, port, out_dir, gB, steps, n_gen):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), ALDM_DIST_BACKEND="gloo")
    from audioldm2_amd import dist as adist
    from audioldm2_amd.pipeline import build_model, seed_everything
    adist.init_distributed()
    torch.manual_seed(7)
    m = build_model(model_name="audioldm2-full").cuda()
    m.scale_factor.fill_(0.75)
    adist.broadcast_module(m, src=0)
    m.clap = _small_clap(0.5)     # half of the CLAP embeddings get replaced by the empty-text one: the decision draws matter
    seed_everything(cases.E2E_SEED)
    m.latent_t_size = 256
    wav = m.generate_batch(_cand_batch(gB), unconditional_guidance_scale=3.5, ddim_steps=steps, n_gen=n_gen, duration=10,
                           shard=(rank, world))
    np.save(os.path.join(out_dir, f"cwave{rank}.npy"), wav)
    torch.save({"sim": m.last_similarity, "best": m.last_best_index, "rng": torch.get_rng_state()},
               os.path.join(out_dir, f"cinfo{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_sharded_run_with_candidates_equals_single_process(tmp_path):
    """VERDICT r2 next #6 / ADVICE r2: prompt sharding with n_candidate_gen_per_text = 2 ON THE GPU — 3 prompts over 2 ranks
    (2 + 1), every rank samples both candidates of its prompts (rows candidate * B + prompt of the global batch, for the
    conditioning and for the noise), re-ranks locally with the CLAP towers at unconditional_prob = 0.5 drawing the GLOBAL
    batch's decisions — and picks, per prompt, the waveform the single-process run picks; the host generator ends in the
    same state."""
    import socket

    import torch.multiprocessing as mp
    from audioldm2_amd.pipeline import build_model, seed_everything
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    gB, steps, n_gen = 3, 4, 2
    mp.spawn(_shard_cand_worker, args=(2, port, str(tmp_path), gB, steps, n_gen), nprocs=2, join=True)
    sharded = np.concatenate([np.load(os.path.join(tmp_path, f"cwave{r}.npy")) for r in range(2)], axis=0)
    infos = [torch.load(os.path.join(tmp_path, f"cinfo{r}.pt")) for r in range(2)]
    torch.manual_seed(7)
    m = build_model(model_name="audioldm2-full").cuda()
    m.scale_factor.fill_(0.75)
    m.clap = _small_clap(0.5)
    seed_everything(cases.E2E_SEED)
    m.latent_t_size = 256
    single = m.generate_batch(_cand_batch(gB), unconditional_guidance_scale=3.5, ddim_steps=steps, n_gen=n_gen, duration=10)
    assert sharded.shape == single.shape == (gB, 1, 163872)
    # the per-prompt choice: rank r's local best index i + c * Bp  <->  global prompt lo + i, candidate c
    chosen = []
    lo = 0
    for r, info in enumerate(infos):
        Bp = len(info["best"])
        chosen += [(lo + (b % Bp), b // Bp) for b in info["best"]]
        lo += Bp
    want = [(b % gB, b // gB) for b in m.last_best_index]
    e = rms(sharded.astype(np.float64) - single) / rms(single)
    report(f"2-rank sharded with {n_gen} candidates vs single process, {gB} prompts: chosen {chosen} vs {want}; wave rel rms {e:.2e}")
    assert chosen == want
    assert e < batch_tol()
    assert all(torch.equal(info["rng"], torch.get_rng_state()) for info in infos)   # same number of host draws everywhere


def _rccl_alone_worker(rank, port, out):
    import torch.distributed as dist
    from audioldm2_amd import dist as adist
    from audioldm2_amd.unet import UNetModel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)   # "nccl" IS RCCL on ROCm
    m = UNetModel(**cases.UNET_TINY).cuda()
    m.load_state_dict(weights.make_state_dict(weights.shapes_of(m), seed=0))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    x, t, ctxs, masks, _ = cases.unet_inputs(cases.UNET_TINY, 2, 16, 8)
    run = lambda: m(x.cuda(), t.cuda(), context_list=[c.cuda() for c in ctxs], context_attn_mask_list=[k.cuda() for k in masks])
    y0 = run()
    sent = adist.broadcast_module(m, src=0, even_alone=True)   # bucketed RCCL broadcasts, then every cache invalidated
    torch.cuda.synchronize()
    same = all(torch.equal(v, before[k]) for k, v in m.state_dict().items())
    y1 = run()
    dist.destroy_process_group()
    torch.save({"sent": sent, "same": same, "equal": bool(torch.equal(y0, y1)), "repacked": m._pk is not None}, out)


@pytest.mark.timeout(600)
def test_rccl_broadcast_module_on_one_gpu(tmp_path):
    """The weight broadcast of dist.py through RCCL itself (backend "nccl"), world size 1 on the one GPU a test box has:
    the library loads, a communicator is created, the bucketed broadcasts launch, the parameters come back unchanged,
    the packed / cached state is dropped and rebuilt, and the forward is bit-identical afterwards.  (N > 1 ranks over
    xGMI need an N-GPU node: the driver's scaling run.)"""
    import socket

    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = os.path.join(tmp_path, "rccl.pt")
    mp.spawn(_rccl_alone_worker, args=(port, out), nprocs=1, join=True)
    r = torch.load(out)
    assert r["sent"] > 0 and r["same"] and r["equal"] and r["repacked"]


def test_ddim_sampling_timesteps_subset_continues_a_full_run(ld):
    """`DDIMSampler.ddim_sampling(timesteps=n)` (ddim.py:198-206): only the first int(min(n / S, 1) * S) - 1 entries of the DDIM
    sequence are sampled.  With eta = 0 the update is deterministic, so restarting from the full run's intermediate state with
    that subset must land on the full run's result."""
    from audioldm2_amd.ddim import DDIMSampler
    from audioldm2_amd.pipeline import seed_everything
    B, S = 1, 8
    batch = cases.e2e_batch(B)
    cond = ld.get_learned_conditioning_dict(batch)
    uncond = {k: ld.cond_stage_models[m["model_idx"]].get_unconditional_condition(B)
              for k, m in ld.cond_stage_model_metadata.items()}
    seed_everything(3)
    s = DDIMSampler(ld)
    s.make_schedule(ddim_num_steps=S, ddim_eta=0.0, verbose=False)
    kw = dict(unconditional_guidance_scale=3.5, unconditional_conditioning=uncond, log_every_t=1)
    full, inter = s.ddim_sampling(cond, (B, 8, 256, 16), **kw)
    assert len(inter["x_inter"]) == S + 1
    n = 5
    subset = int(min(n / S, 1) * S) - 1                      # 4 steps: sequence entries 3, 2, 1, 0
    start = inter["x_inter"][S - subset]                     # the state after the first S - subset steps of the full run
    part, _ = s.ddim_sampling(cond, (B, 8, 256, 16), x_T=start, timesteps=n, **kw)
    assert rel(part, full) < 1e-5
    with pytest.raises(NotImplementedError):
        s.ddim_sampling(cond, (B, 8, 256, 16), ddim_use_original_steps=True, **kw)
    # DDIMSampler.decode (ddim.py:452-491): the last `t_start` steps from a given state, one eager p_sample_ddim per step —
    # the same continuation through the other entry point
    dec = s.decode(start.clone(), cond, subset, unconditional_guidance_scale=3.5, unconditional_conditioning=uncond)
    assert rel(dec, full) < 1e-5
    # stochastic_encode (ddim.py:434-449) against its closed form on the schedule's tables
    x0 = torch.randn(B, 8, 256, 16, generator=torch.Generator().manual_seed(5)).cuda()
    nz = torch.randn(B, 8, 256, 16, generator=torch.Generator().manual_seed(6))
    t = torch.tensor([3] * B)
    enc = s.stochastic_encode(x0, t, noise=nz)
    want = float(s.ddim_alphas[3]) ** 0.5 * x0.cpu() + float(s.ddim_sqrt_one_minus_alphas[3]) * nz
    assert rel(enc.cpu(), want) < 1e-6
