"""GPU parity of the remaining BASELINE configurations (configs[3], configs[4]) end to end against fixtures produced by
the REAL reference (oracle/make_golden.py e2espeech / e2elarge; the CPU oracle is pinned to the same fixtures in
tests/test_oracle.py).  Same tolerances as tests/test_model_gpu.py; kept in its own module so it runs after the op and
model suites."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cases, weights

from tolerances import fused_tol, latent_tol, log_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt((a ** 2).mean()))


def report(line):
    print(line)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_report.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("model_name,fixture,keys_json,B,steps,mode", [
    ("audioldm2-speech-gigaspeech", "e2e_speech_2step_b1", "e2espeech_statedict_keys.json", 1, 2, None),
    ("audioldm2-full-large-1150k", "e2e_large_2step_b1", "e2elarge_statedict_keys.json", 1, 2, None),
    ("audioldm2-speech-gigaspeech", "e2e_speech_20step_b2", "e2espeech_statedict_keys.json", 2, 20, None),
    ("audioldm2-full-large-1150k", "e2e_large_20step_b2", "e2elarge_statedict_keys.json", 2, 20, None),
    ("audioldm2-speech-gigaspeech", "e2e_speech_5step_b8", "e2espeech_statedict_keys.json", 8, 5, "bf16x6"),
    ("audioldm2-speech-gigaspeech", "e2e_speech_5step_b8", "e2espeech_statedict_keys.json", 8, 5, "bf16x3"),
    ("audioldm2-full-large-1150k", "e2e_large_5step_b8", "e2elarge_statedict_keys.json", 8, 5, "bf16x6"),
    ("audioldm2-full-large-1150k", "e2e_large_5step_b8", "e2elarge_statedict_keys.json", 8, 5, "bf16x3"),
    ("audioldm2-speech-gigaspeech", "e2e_speech_5step_b8", "e2espeech_statedict_keys.json", 8, 5, "f16x3"),
    ("audioldm2-full-large-1150k", "e2e_large_5step_b8", "e2elarge_statedict_keys.json", 8, 5, "f16x3")])
def test_e2e_speech_and_large_match_reference_generate_batch(model_name, fixture, keys_json, B, steps, mode):
    """BASELINE configs 4 / 5 end to end against the REAL reference's generate_batch fixtures (B=1 at 2 DDIM steps, B=2 at
    20 steps, and — VERDICT r3 next #1b — B=8 at 5 steps = the batch bench.py's `configs` numbers are measured at, in both
    product modes, so the kernel instantiations the tuned tables pick for 16-sample passes are the ones compared with the
    reference; CFG 3.5, seed 42): the speech model's single 512-token context (masked cross attention over 512 keys) and the
    large model's three context slots with transformer depth 2.  Random-init weights; the waveform error is asserted against
    the distance between two unrelated samples' waveforms (fixture `wave_between_rms`), not only against 1e-3."""
    from audioldm2_amd import ops
    if not os.path.exists(os.path.join(GOLD, fixture + ".npz")):
        pytest.skip(f"{fixture} absent")
    prev = ops.set_mma(mode) if mode else None
    try:
        _e2e_named(model_name, fixture, keys_json, B, steps, mode or ops.MMA_MODE)
    finally:
        if prev:
            ops.set_mma(prev)


def _e2e_named(model_name, fixture, keys_json, B, steps, mode):
    from audioldm2_amd.pipeline import build_model, seed_everything
    g = gold(fixture)
    m = build_model(model_name=model_name)
    with open(os.path.join(GOLD, keys_json)) as f:
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
    m.latent_t_size = 256
    wave = m.generate_batch(cases.e2e_batch(B), unconditional_guidance_scale=3.5, ddim_steps=steps, n_gen=1, duration=10)
    assert wave.shape == (B, 1, int(g["wave_len"]))
    el = rms(rec["latent"].double().cpu().numpy() - g["latent"]) / rms(g["latent"])
    eh = rms(wave[..., :32768].astype(np.float64) - g["wave_head"])
    ed = rms(wave[..., ::16].astype(np.float64) - g["wave_dec"])
    between = float(g["wave_between_rms"])
    report(f"{model_name} e2e {steps} steps B={B} [{mode}]: latent rel rms {el:.2e}  wave(head) rms_err {eh:.3e}  wave(1/16) rms_err "
           f"{ed:.3e} / rms_ref {float(g['wave_rms']):.3e} / between-sample {between:.3e}")
    assert log_err(el, latent_tol(5, mode), f"{model_name} latent {steps} steps B={B}") < latent_tol(5, mode)   # per-mode bar, <= 5x measured
    assert between > 1e-2 and max(eh, ed) < 1e-3 and max(eh, ed) < 1e-3 * between
    del m
    torch.cuda.empty_cache()


def _ref_attention(q, k, v, heads, mask=None):
    """attention.py:343-367 restated (einsum / masked_fill(-finfo.max) / softmax / einsum) in the dtype of its arguments."""
    B, Lq, Cc = q.shape
    d = Cc // heads

    def sp(t):
        return t.view(B, -1, heads, d).permute(0, 2, 1, 3).reshape(B * heads, -1, d)
    qh, kh, vh = sp(q), sp(k), sp(v)
    sim = torch.einsum("bid,bjd->bij", qh, kh) * d ** -0.5
    if mask is not None:
        m = mask.reshape(B, -1)[:, None, :].repeat_interleave(heads, 0)
        sim = sim.masked_fill(~(m == 1), -torch.finfo(sim.dtype).max)
    o = torch.einsum("bij,bjd->bid", sim.softmax(-1), vh)
    return o.view(B, heads, Lq, d).permute(0, 2, 1, 3).reshape(B, Lq, Cc)


def _attention_case(Lk, masked, mode):
    from audioldm2_amd import ops
    B, heads, Lq = 16, 8, 1024  # BASELINE batch: cdiv(Lq, 256) * heads * B = 512 blocks -> 64 queries per wave
    Cc = heads * 32
    gen = torch.Generator().manual_seed(7)
    q = torch.randn(B, Lq, Cc, generator=gen)
    kv = torch.randn(B, Lk, 2 * Cc, generator=gen)
    mask = None
    if masked:
        mask = (torch.rand(B, Lk, generator=gen) > 0.25).float()
        mask[:, 0] = 1
    ref = _ref_attention(q.double(), kv[:, :, :Cc].double().contiguous(), kv[:, :, Cc:].double().contiguous(), heads, mask)   # fp64
    kvd = kv.cuda()
    prev = ops.attention_mma(mode)
    try:
        y = ops.attention(q.cuda(), kvd[:, :, :Cc], kvd[:, :, Cc:], heads, mask=None if mask is None else mask.cuda())
    finally:
        ops.attention_mma(prev)
    return float((y.double().cpu() - ref.double()).abs().max() / ref.double().abs().max())


@pytest.mark.parametrize("Lk,masked", [(1024, False), (512, True)])
def test_attention_64_queries_per_wave_instantiations(Lk, masked):
    """aldm_attention_d32 at the BASELINE batch runs its QT = 2 instantiations (64 queries per wave), which the op tests'
    small batches never reach: self attention 1024 x 1024 (every UNet config) and the speech model's masked cross
    attention over 512 keys.  Max-norm relative error <= 5e-6 vs the reference's einsum/softmax/einsum evaluated in fp64."""
    e = _attention_case(Lk, masked, 1)
    report(f"attention d32, 16 x 8 heads x 1024 queries x {Lk} keys{' masked' if masked else ''}, QT=2: rel err {e:.2e}")
    assert log_err(e, fused_tol("f32"), "attention QT=2 f32") < fused_tol("f32")


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("Lk,masked", [(1024, False), (512, True)])
def test_attention_bf16_split_64_queries_per_wave_instantiations(Lk, masked, mode):
    """The same two shapes on the bf16 matrix cores: bf16x6 (mode 2) and the default bf16x3 (mode 3) — full key tiles, so the
    software-pipelined kernel with 64 queries per wave, without and with a key mask.  Per-mode bars: 5e-6 / 5e-5."""
    name = {2: "bf16x6", 3: "bf16x3"}[mode]
    e = _attention_case(Lk, masked, mode)
    report(f"attention d32 [{name}], 16 x 8 heads x 1024 queries x {Lk} keys{' masked' if masked else ''}, QT=2: rel err {e:.2e}")
    assert log_err(e, fused_tol(name), f"attention QT=2 {name}") < fused_tol(name)
