"""CPU tests of the host-side mirror of the reference interface (no kernels): state-dict
compatibility with the reference (key names + shapes recorded from the real classes), DDIM schedule
tables, config dicts, conditioning routing, RNG draw order."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _keys(fname, key=None):
    with open(os.path.join(GOLD, fname)) as f:
        d = json.load(f)
    d = d if key is None else d[key]
    return {k: tuple(v) for k, v in d.items()}


def _shapes(m):
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


@pytest.mark.parametrize("name,cfg", [("unet_tiny", cases.UNET_TINY), ("unet_large_tiny", cases.UNET_LARGE_TINY),
                                      ("unet_film_tiny", cases.UNET_FILM_TINY), ("unet_full", cases.UNET_FULL)])
def test_unet_state_dict_matches_reference(name, cfg):
    from audioldm2_amd.unet import UNetModel
    assert _shapes(UNetModel(**cfg)) == _keys("unet_statedict_keys.json", name)


@pytest.mark.parametrize("name,dd", [("vae16k", cases.DDCONFIG_16K), ("vae48k", cases.DDCONFIG_48K)])
def test_autoencoder_state_dict_matches_reference(name, dd):
    """encoder.*, decoder.*, quant_conv.*, post_quant_conv.*, vocoder.* (398 tensors for 16k)."""
    from audioldm2_amd.vae import AutoencoderKL
    ae = AutoencoderKL(ddconfig=dd, embed_dim=dd["z_channels"], image_key="fbank")
    assert _shapes(ae) == _keys("vae_statedict_keys.json", name)


def test_latent_diffusion_hot_path_keys_and_reference_checkpoint_loading():
    """`model.diffusion_model.*` (1520) + `first_stage_model.*` (398) as in the real LatentDiffusion;
    load_reference_state_dict tolerates cond_stage_models.* / model_ema.* / clap.* entries."""
    from audioldm2_amd.pipeline import build_model
    ld = build_model(model_name="audioldm2-full")
    ref = _keys("e2e_statedict_keys.json")
    mine = {k: v for k, v in _shapes(ld).items() if k.startswith("model.diffusion_model.") or k.startswith("first_stage_model.")}
    assert mine == ref
    assert len([k for k in ref if k.startswith("model.diffusion_model.")]) == 1520
    sd = {k: torch.zeros(v) for k, v in _shapes(ld).items()}
    sd["model_ema.decay"] = torch.zeros(())
    sd["cond_stage_models.0.x"] = torch.zeros(3)
    sd["scale_factor"] = torch.tensor(0.33)
    ignored = ld.load_reference_state_dict(sd)
    assert "model_ema.decay" in ignored and float(ld.scale_factor) == pytest.approx(0.33)
    del sd["first_stage_model.decoder.conv_in.weight"]
    with pytest.raises(RuntimeError, match="lacks"):
        ld.load_reference_state_dict(sd)


@pytest.mark.parametrize("model_name,keys_json", [("audioldm_48k", "e2e48k_statedict_keys.json"),
                                                  ("audioldm2-speech-gigaspeech", "e2espeech_statedict_keys.json"),
                                                  ("audioldm2-full-large-1150k", "e2elarge_statedict_keys.json")])
def test_other_configs_hot_path_keys_match_the_reference(model_name, keys_json):
    """BASELINE configs 3-5: the hot-path tensors of build_model(name) are exactly the real reference's
    (names and shapes recorded by oracle/make_golden.py from the reference's LatentDiffusion)."""
    from audioldm2_amd.pipeline import build_model
    m = build_model(model_name=model_name)
    ours = {k: tuple(v.shape) for k, v in m.state_dict().items()
            if k.startswith("model.diffusion_model.") or k.startswith("first_stage_model.")}
    with open(os.path.join(GOLD, keys_json)) as f:
        ref = {k: tuple(v) for k, v in json.load(f).items()}
    assert ours == ref


def test_ddim_sampler_tables_match_reference_exactly():
    from audioldm2_amd.ddim import DDIMSampler
    from audioldm2_amd.pipeline import build_model
    g = np.load(os.path.join(GOLD, "ddim_tables.npz"))

    class M:
        num_timesteps = 1000
    m = M()
    from oracle.ddim import make_schedule_buffers
    m.alphas_cumprod = make_schedule_buffers()["alphas_cumprod"]
    for S, eta in [(200, 1.0), (50, 0.0), (5, 1.0)]:
        s = DDIMSampler(m)
        s.make_schedule(S, ddim_eta=eta, verbose=False)
        assert np.array_equal(s.ddim_timesteps, g[f"ts_{S}"])
        assert np.array_equal(s.ddim_alphas.numpy(), g[f"alphas_{S}"])
        assert np.array_equal(np.asarray(s.ddim_alphas_prev), g[f"alphas_prev_{S}"])
        assert np.array_equal(s.ddim_sigmas.numpy(), g[f"sigmas_{S}"])
        assert np.array_equal(s.ddim_sqrt_one_minus_alphas, g[f"som_{S}"])
        from oracle.ddim import ddim_tables
        assert torch.equal(s.ddim_coef, ddim_tables(m.alphas_cumprod, S, eta)[1])
    # the module's own schedule buffers equal the oracle's restatement of ddpm.py:201-303
    ld_buf = build_model.__globals__["LatentDiffusion"].register_schedule
    assert callable(ld_buf)


def test_noise_draw_order_is_the_reference_order():
    """x_T then one randn per step from the global CPU generator (ddim.py:191,351)."""
    from audioldm2_amd.ddim import DDIMSampler

    class M:
        num_timesteps = 1000
    s = DDIMSampler(M())
    shape = (2, 8, 4, 4)
    torch.manual_seed(7)
    img, noise, _ = s._draw_noise(shape, 3, None, False)
    torch.manual_seed(7)
    ref = [torch.randn(shape) for _ in range(4)]
    assert torch.equal(img, ref[0]) and all(torch.equal(noise[i], ref[i + 1]) for i in range(3))
    torch.manual_seed(7)
    img, noise, q = s._draw_noise(shape, 2, None, True)  # inpainting: q_sample noise precedes step noise
    torch.manual_seed(7)
    ref = [torch.randn(shape) for _ in range(5)]
    assert torch.equal(q[0], ref[1]) and torch.equal(noise[0], ref[2]) and torch.equal(q[1], ref[3])


def test_config_and_conditioning_routing():
    from audioldm2_amd.pipeline import DiffusionWrapper, FixedCond, default_audioldm_config
    for name, ctx, depth in [("audioldm2-full", [768, 1024], 1), ("audioldm2-full-large-1150k", [768, 1024, None], 2),
                             ("audioldm2-speech-gigaspeech", [768], 1), ("audioldm_48k", [None], 1)]:
        p = default_audioldm_config(name)["model"]["params"]
        assert p["unet_config"]["params"]["context_dim"] == ctx
        assert p["unet_config"]["params"]["transformer_depth"] == depth
    p = default_audioldm_config("audioldm_48k")["model"]["params"]
    assert p["latent_t_size"] == 128 and p["latent_f_size"] == 32 and p["channels"] == 16
    a = FixedCond("crossattn", 768, 8, uncond_zero=True, device="cpu")
    b = FixedCond("crossattn", 1024, 32, uncond_length=1, masked_tail=8, device="cpu")
    batch = cases.e2e_batch(4)
    cond = {"crossattn_audiomae_generated": a(batch), "crossattn_flan_t5": b(batch["text"])}
    y, ctxs, masks = DiffusionWrapper.route(cond)
    assert y is None and [tuple(c.shape) for c in ctxs] == [(4, 8, 768), (4, 32, 1024)]
    assert masks[1][1, -8:].sum() == 0 and masks[1][0].sum() == 32
    u = b.get_unconditional_condition(4)
    assert tuple(u[0].shape) == (4, 1, 1024) and float(a.get_unconditional_condition(2)[0].abs().max()) == 0.0
    f = FixedCond("film", 512, device="cpu")
    y, ctxs, _ = DiffusionWrapper.route({"film_clap_cond1": f(batch["text"])})
    assert tuple(y.shape) == (4, 512) and ctxs == []


def test_igemm_tuning_table_is_wellformed():
    """audioldm2_amd/tuning/mi355x_igemm.json (measured launch configurations): keys follow
    ops._TUNE_FIELDS + pre_mode, values name supported tiles; ops.tune_key spells keys the same way."""
    import json
    from audioldm2_amd import lib, ops
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audioldm2_amd", "tuning",
                        "mi355x_igemm.json")
    with open(path) as f:
        t = json.load(f)
    assert t["fields"] == list(ops._TUNE_FIELDS) + ["pre_mode"]
    tiles = {(128, 128), (128, 64), (64, 128), (64, 64), (128, 32)}
    assert len(t["entries"]) > 0
    for k, v in t["entries"].items():
        parts = k.split(",")
        assert len(parts) == len(t["fields"]) and all(p.lstrip("-").isdigit() for p in parts)
        assert (v[0], v[1]) in tiles and 1 <= v[2] <= 16 and v[3] in (1, 2)
    d = lib.IgemmDesc()
    d.B, d.H, d.W, d.C1, d.KH, d.KW, d.N = 1, 1, 77, 64, 1, 1, 32
    assert len(ops.tune_key(d).split(",")) == len(t["fields"])
    # the bf16-split table carries one more value per entry: which matrix-core path won (1 = fp32 MFMA)
    with open(path.replace("mi355x_igemm.json", "mi355x_igemm_bf16x6.json")) as f:
        tb = json.load(f)
    assert tb["fields"] == t["fields"] and tb["mma"] == "bf16x6" and len(tb["entries"]) > 0
    for k, v in tb["entries"].items():
        assert len(k.split(",")) == len(tb["fields"])
        assert (v[0], v[1]) in tiles and 1 <= v[2] <= 16 and v[3] in (1, 2) and v[4] in (0, 1)
    assert ops._tuned_table(True) and all(len(v) == 5 for v in ops._tuned_table(True).values())
    assert all(len(v) == 4 for v in ops._tuned_table(False).values())


def test_packed_weight_split_image_is_lazy_and_mode_gated(monkeypatch):
    """Packed.split_ptr(): no split image (and no library call) on the fp32-MFMA path; set_mma validates."""
    import torch
    from audioldm2_amd import ops
    pw = ops.Packed(torch.zeros(4), N=1, Cin=4, KH=1, KW=1)
    prev = ops.set_mma("f32")
    try:
        assert pw.split_ptr() is None and pw.split is None
        with pytest.raises(AssertionError):
            ops.set_mma("bf16")
    finally:
        ops.set_mma(prev)


def test_bench_contract_defaults_and_no_cpu_path():
    """bench.py: flag defaults are the BASELINE configuration (N = 1, batch 8, 200 DDIM steps, a K/W that finishes in
    minutes), the roofline peaks are the guide's numbers, and without a GPU it refuses to run (no CPU fallback)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    old = sys.argv
    sys.argv = ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = old
    assert (a.gpus, a.steps, a.warmup, a.batch, a.ddim_steps, a.model, a.mma) == (1, 2, 1, 8, 200, "audioldm2-full", None)
    assert bench.PEAK_F32_MFMA_TFLOPS == 157.3 and bench.PEAK_BF16X6_TFLOPS == 416.7
    assert set(bench.UNET_GFLOP_PER_FWD_SAMPLE) == {"audioldm2-full", "audioldm2-full-large-1150k",
                                                    "audioldm2-speech-gigaspeech", "audioldm_48k"}
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "needs the MI355X" in r.stderr and not r.stdout.strip()


@pytest.mark.timeout(600)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) re-launches itself under torch.distributed.run with
    a 127.0.0.1 rendezvous.  --dry-launch runs the launcher and the pre-kernel distributed set-up on gloo:
    two ranks, contiguous prompt slices of the global batch, the weight broadcast reaching both; without --dry-launch the
    preflight refuses a node that has fewer GPUs than ranks instead of dying in an assert inside a rank."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True,
                       text=True, timeout=500, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout   # ONE line, from rank 0
    rec = json.loads(lines[0])
    assert rec["dry_launch"] and rec["n_gpus"] == 2 and rec["global_batch"] == 16 and rec["launched_by"] == "torch.distributed.run"
    assert [e["rank"] for e in rec["ranks"]] == [0, 1] and [e["prompts"] for e in rec["ranks"]] == [[0, 8], [8, 16]]
    assert len({e["pid"] for e in rec["ranks"]}) == 2 and rec["weight_broadcast_bytes"] == (32 * 32 + 32) * 4
    assert len({e["weight_checksum"] for e in rec["ranks"]}) == 1
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 2 and "exposes 0 GPU(s)" in r.stderr and not r.stdout.strip()


def _torch_ops_stand_in():
    """Torch (CPU) stand-ins for the handful of ops audioldm2_amd/seqgen.py calls — TEST ONLY: lets the host logic of the
    sequence generator (cache bookkeeping, masks, positions, head split / merge) run on a CPU-only box against the
    reference fixtures.  The product has no such path: the real ops raise on CPU tensors."""
    import math
    import types
    import torch.nn.functional as F
    from audioldm2_amd.lib import ACT_GELU_TANH

    class PW:
        def __init__(self, w, b):
            self.w, self.b = w.detach().float(), None if b is None else b.detach().float()

    def linear(x, pw, act=0, res=None):
        y = F.linear(x, pw.w, pw.b)
        if act == ACT_GELU_TANH:
            y = 0.5 * y * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (y + 0.044715 * y ** 3)))
        else:
            assert act == 0
        return y if res is None else y + res

    def softmax_rows_masked(x, keymask, q_pos0, scale=1.0):
        B, H, T, N = x.shape
        j = torch.arange(N)[None, :]
        ok = (keymask[:, None, None, :] != 0) & (j <= (q_pos0 + torch.arange(T))[:, None])[None, None]
        return torch.where(ok, x * scale, torch.full([], float("-inf"))).softmax(-1)

    def _act(y, act):
        if act == ACT_GELU_TANH:
            return 0.5 * y * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (y + 0.044715 * y ** 3)))
        assert act == 0
        return y

    def decode_linear(x, w_kn, bias=None, *, ln=None, act=0, res=None):
        if ln is not None:
            x = F.layer_norm(x, (x.shape[-1],), ln[0], ln[1], ln[2])
        y = _act(x @ w_kn + (0 if bias is None else bias), act)
        return y if res is None else y + res

    def decode_attention(qkv, pos, kc, vc, keymask, heads):
        B, E = qkv.shape[0], heads * 64
        n_tot = keymask.shape[1]
        q, k, v = (qkv[:, i * E:(i + 1) * E].reshape(B, heads, 1, 64) for i in range(3))
        kc.view(B, heads, n_tot, 64).index_copy_(2, pos, k)
        vc.view(B, heads, n_tot, 64).index_copy_(2, pos, v)
        s = (q.reshape(B * heads, 1, 64) @ kc.transpose(1, 2)) * 0.125
        s = torch.where(keymask[:, None, None, :] != 0, s.view(B, heads, 1, n_tot), torch.full([], float("-inf")))
        return (s.softmax(-1).view(B * heads, 1, n_tot) @ vc).view(B, E)

    # the split-K decode step (round 6): partial slabs [S, M, N] handed from launch to launch (S = 3 here: the stand-in slices K
    # into thirds, so the hand-over logic — who adds which bias, where the activation sits — is what gets exercised)
    def decode_gemv(x, w_kn, *, xbias=None, xact=0):
        xe = x if x.dim() == 2 else x.sum(0)
        xe = _act(xe + (0 if xbias is None else xbias), xact)
        K = w_kn.shape[0]
        return torch.stack([xe[:, i * K // 3:(i + 1) * K // 3] @ w_kn[i * K // 3:(i + 1) * K // 3] for i in range(3)])

    def decode_reduce_ln(part, *, bias=None, bias_row=None, res=None, ln=None, want_h=True, xn_out=None):
        h = 0 if part is None else part.sum(0)
        if bias is not None:
            h = h + (bias[int(bias_row)] if bias_row is not None else bias)
        if res is not None:
            h = h + res
        if ln is None:
            return h
        xn = F.layer_norm(h, (h.shape[-1],), ln[0], ln[1], ln[2])
        if xn_out is not None:
            xn_out.copy_(xn)
        return (h, xn) if want_h else xn

    def decode_attention_parts(qkv_part, qbias, pos, kc, vc, keymask, heads):
        return decode_attention(qkv_part.sum(0) + (0 if qbias is None else qbias), pos, kc, vc, keymask, heads)

    return types.SimpleNamespace(
        DECODE_MAX_ROWS=16, decode_linear=decode_linear, decode_attention=decode_attention,
        decode_gemv=decode_gemv, decode_reduce_ln=decode_reduce_ln, decode_attention_parts=decode_attention_parts,
        pack_conv=lambda w, b=None: PW(w, b), linear=linear,
        layernorm=lambda x, g, b, eps=1e-5: F.layer_norm(x, (x.shape[-1],), g, b, eps),
        axpby=lambda a, b, alpha, beta=0.0: alpha * a + beta * b,
        gemm_nt=lambda a, bm, alpha=1.0: alpha * a @ bm.transpose(1, 2),
        softmax_rows_masked=softmax_rows_masked, pack_kn=lambda src: src,
        gemm_packed_batched=lambda a, bp, K, N: a @ bp)


@pytest.mark.parametrize("decode", ["split", "fast", "general"])
@pytest.mark.parametrize("fixture,cfg_name,T", [("seqgen_full_8step_b2", "SEQGEN_FULL", 20),
                                                ("seqgen_speech_24step_b2", "SEQGEN_SPEECH", 40)])
def test_sequence_generator_host_logic_matches_reference_fixture(monkeypatch, fixture, cfg_name, T, decode):
    """audioldm2_amd.seqgen.Sequence2AudioMAE: the reference's state-dict keys load strictly, and its key/value-cached
    decode (fixed-length cache, masked future positions) reproduces the REAL reference's generate() fixture when the
    device ops are replaced by torch stand-ins — i.e. the orchestration is right; the kernels are tests/test_seqgen_gpu.py's
    business."""
    from audioldm2_amd import seqgen
    from oracle import cases, weights
    cfg = getattr(cases, cfg_name)
    monkeypatch.setattr(seqgen, "ops", _torch_ops_stand_in())
    monkeypatch.setenv("ALDM_SEQGEN_DECODE", decode)   # the decode step (>= 16 tokens) on the single-position ops / the general ones
    m = seqgen.Sequence2AudioMAE(base_learning_rate=2e-4, sequence_gen_length=cfg["steps"], sequence_input_key=cfg["keys"],
                                 sequence_input_embed_dim=cfg["dims"], cond_stage_config={}, batchsize=16)
    with open(os.path.join(GOLD, fixture + "_keys.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    ours = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert {k: v for k, v in ours.items() if k != "model.wte.weight"} == shapes and ours["model.wte.weight"] == (50257, 768)
    m.load_state_dict(weights.make_state_dict(shapes, seed=0), strict=False)
    out, _ = m.generate(None, cond_dict=cases.seqgen_cond(cfg, 2, T))
    want = torch.from_numpy(np.load(os.path.join(GOLD, fixture + ".npz"))["out"])
    assert out.shape == want.shape
    assert float((out - want).abs().max() / want.abs().max()) < 1e-5


def _torch_ops_stand_in_clap():
    """Torch (CPU) stand-ins for the ops audioldm2_amd/clap.py calls (both towers) — TEST ONLY, like `_torch_ops_stand_in`:
    what is checked is the module's own logic (window / shift / merge index tables, the bias + shift-mask assembly, head
    split and merge, the log-mel and BatchNorm folding, the draw order of the unconditional replacement), against the
    fixtures generated with the REAL reference classes."""
    import math
    import types
    import torch.nn.functional as F
    from audioldm2_amd.lib import ACT_GELU, ACT_LOGCLAMP, ACT_LRELU, ACT_TANH
    base = _torch_ops_stand_in()

    def linear(x, pw, act=0, res=None, act_slope=0.0):
        y = F.linear(x, pw.w, pw.b)
        if act == ACT_LOGCLAMP:
            y = torch.log(torch.clamp(y, min=act_slope))
        elif act == ACT_GELU:
            y = F.gelu(y)
        elif act == ACT_LRELU:
            y = F.leaky_relu(y, act_slope)
        elif act == ACT_TANH:
            y = torch.tanh(y)
        else:
            assert act == 0, act
        return y if res is None else y + res

    def reflect_pad_1d(x, pad):
        y = F.pad(x[:, None], (pad, pad), mode="reflect")[:, 0]
        return F.pad(y, (0, (-y.shape[1]) % 4))

    def frames_gemm(sig, frames, hop, pw):
        return sig.unfold(1, pw.w.shape[1], hop)[:, :frames] @ pw.w.t()

    def power_spec(spec, Fq, ld_out):
        out = torch.zeros(spec.shape[0], ld_out)
        out[:, :Fq] = spec[:, :Fq] ** 2 + spec[:, Fq:2 * Fq] ** 2
        return out

    def bicubic_patchify(x, S, p):
        B, T, Fm = x.shape
        ratio = S // Fm
        img = x[:, None]
        if T < S * ratio:
            img = F.interpolate(img, (S * ratio, Fm), mode="bicubic", align_corners=True)
        img = img.permute(0, 1, 3, 2).reshape(B, 1, Fm, ratio, S).permute(0, 1, 3, 2, 4).reshape(B, 1, S, S)
        return F.unfold(img, kernel_size=p, stride=p).transpose(1, 2).contiguous()

    def softmax_rows_bias(x, bias, keymask, scale=1.0):
        s = x * scale + bias[None]
        return torch.where(keymask[:, None, None, :] != 0, s, torch.full([], float("-inf"))).softmax(-1)

    def resample_sinc(x, k, down, up, width, out_len):
        y = F.conv1d(F.pad(x[:, None], (width, width + down)), k[:, None], stride=down)
        return y.transpose(1, 2).reshape(x.shape[0], -1)[:, :out_len].contiguous()

    def rowscale_add(x, s, res=None, divide=False):
        y = x / torch.clamp(s, min=1e-12)[:, None] if divide else x * s[:, None]
        return y if res is None else y + res

    return types.SimpleNamespace(
        **{**vars(base), "linear": linear, "reflect_pad_1d": reflect_pad_1d, "frames_gemm": frames_gemm,
           "power_spec": power_spec, "col_affine": lambda x, sc, sh: x * sc + sh, "bicubic_patchify": bicubic_patchify,
           "softmax_rows_bias": softmax_rows_bias, "token_mean": lambda x: x.mean(1),
           "row_l2norm": lambda x, Fq: x[:, :Fq].norm(dim=-1), "rowscale_add": rowscale_add, "resample_sinc": resample_sinc,
           "row_cosine": lambda a, b, eps=1e-8: F.cosine_similarity(a, b, dim=-1, eps=eps),
           "ACT_GELU": ACT_GELU, "ACT_LOGCLAMP": ACT_LOGCLAMP, "ACT_LRELU": ACT_LRELU, "ACT_TANH": ACT_TANH})


def test_clap_towers_host_logic_matches_reference_fixtures(monkeypatch):
    """audioldm2_amd.clap, audio AND text mode, with the device ops replaced by torch stand-ins on the CPU: the HTSAT forward
    (front end, window partition / cyclic shift / patch merging as row gathers, relative-position bias + shift mask,
    pooling, projection) reproduces the fixture generated with the REAL `HTSAT_Swin_Transformer`, the RoBERTa tower the
    transformers fixture, and `cos_similarity` draws the unconditional replacements in the reference's order."""
    from audioldm2_amd import clap
    from oracle import cases, weights
    monkeypatch.setattr(clap, "ops", _torch_ops_stand_in_clap())
    monkeypatch.setattr(clap, "_DEV", torch.device("cpu"))
    with open(os.path.join(GOLD, "htsat_keys.json")) as f:
        ashapes = {k: tuple(v) for k, v in json.load(f).items()}
    with open(os.path.join(GOLD, "clap_text_keys.json")) as f:
        tshapes = {k: tuple(v) for k, v in json.load(f).items()}
    m = clap.CLAPAudioEmbeddingClassifierFreev2(embed_mode="audio", unconditional_prob=0.0, sampling_rate=16000,
                                                config=cases.clap_text_test_config(), audio_config=cases.htsat_test_config())
    sd = {**cases.htsat_state_dict(ashapes), **weights.make_state_dict(tshapes, seed=0)}
    missing = m.model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    ids, mask = cases.clap_text_tokens()
    m.build_unconditional_emb({"input_ids": ids[2:3].repeat(2, 1), "attention_mask": mask[2:3].repeat(2, 1)})
    ga = np.load(os.path.join(GOLD, "htsat_base2222_b2.npz"))
    gt = np.load(os.path.join(GOLD, "clap_text_base2_b3.npz"))
    wav = cases.clap_waveform(2)
    out = m(wav[:, None])                                   # forward(), "audio" mode
    want = torch.from_numpy(ga["emb"])
    assert tuple(out.shape) == (2, 1, 512)
    assert float((out[:, 0] - want).abs().max() / want.abs().max()) < 2e-5
    m.embed_mode = "text"
    temb = m.encode_tokens(ids, mask)
    assert float((temb[:, 0] - torch.from_numpy(gt["emb"])).abs().max()) < 2e-5
    m.embed_mode = "audio"
    # cos_similarity: audio draws first, then text (encoders/modules.py:639-653); rows replaced by the unconditional token
    m.unconditional_prob = 0.5
    torch.manual_seed(3)
    sim = m.cos_similarity(wav, {"input_ids": ids[:2], "attention_mask": mask[:2]})
    torch.manual_seed(3)
    draws = [float(torch.rand(1)) < 0.5 for _ in range(4)]
    u = m.unconditional_token[0]
    a2 = torch.stack([u if draws[i] else want[i] for i in range(2)])
    t2 = torch.stack([u if draws[2 + i] else torch.from_numpy(gt["emb"])[i] for i in range(2)])
    ref = torch.nn.functional.cosine_similarity(a2, t2, dim=-1)
    assert torch.allclose(sim, ref, atol=5e-5) and m.embed_mode == "audio"


def test_t5_and_phoneme_host_logic_match_reference_fixtures(monkeypatch):
    """audioldm2_amd.t5 / audioldm2_amd.phoneme with the device ops replaced by torch stand-ins on the CPU: the bucketed
    relative-position bias, key padding to a multiple of 4, the fused q/k/v and gated-FF weight layouts (T5); masks, the
    scaled embedding table, the FFN's masked convolutions and the positional embedding (phoneme encoder) reproduce the
    fixtures generated with the REAL reference classes."""
    import math
    import types
    import torch.nn.functional as F
    from audioldm2_amd import phoneme as pph
    from audioldm2_amd import t5 as pt5
    from audioldm2_amd.lib import ACT_GELU_TANH, ACT_LRELU
    from oracle import cases
    from oracle import phoneme as oph
    base = _torch_ops_stand_in_clap()

    def linear_geglu(x, pw, split_out=None, gate_act=0):
        y = F.linear(x, pw.w, pw.b)
        inner = y.shape[-1] // 2
        gate = y[..., inner:]
        assert gate_act == ACT_GELU_TANH
        return y[..., :inner] * (0.5 * gate * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (gate + 0.044715 * gate ** 3))))

    def conv(x, pw, pad=(0, 0), act=0, act_slope=0.0):   # the phoneme FFN: [B, 1, T, C] * [N, C, k] along T
        y = F.conv1d(x[:, 0].transpose(1, 2), pw.w, pw.b, padding=pad[1]).transpose(1, 2)
        if act == ACT_LRELU:
            y = F.leaky_relu(y, act_slope)
        return y[:, None]

    def rel_attention(q, k, v, heads, ek, ev, mask):
        B, T, C = q.shape
        hs = lambda t: t.reshape(B, T, heads, C // heads).transpose(1, 2)
        return oph.rel_attention(hs(q), hs(k), hs(v), ek[None], ev[None], mask).transpose(1, 2).reshape(B, T, C)

    def rowscale_add(x, s, res=None, divide=False):
        y = x * s.reshape(*x.shape[:-1], 1)
        return y if res is None else y + res

    ops_t = types.SimpleNamespace(**{**vars(base), "pack_geglu": base.pack_conv, "linear_geglu": linear_geglu,
                                     "rmsnorm": lambda x, w, eps=1e-6: w * x * torch.rsqrt((x * x).mean(-1, keepdim=True) + eps),
                                     "conv": conv, "rel_attention": rel_attention, "rowscale_add": rowscale_add,
                                     "ACT_GELU_TANH": ACT_GELU_TANH})
    cpu = torch.device("cpu")
    # ---- FLAN-T5
    monkeypatch.setattr(pt5, "ops", ops_t)
    monkeypatch.setattr(pt5, "_DEV", cpu)
    with open(os.path.join(GOLD, "t5_keys.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    m = pt5.FlanT5HiddenState(config=cases.t5_test_config())
    m.model.load_state_dict(cases.t5_state_dict(shapes), strict=True)
    g = np.load(os.path.join(GOLD, "t5_large3_b3.npz"))
    ids, mask = cases.t5_tokens()
    h, am = m.encode_tokens(ids, mask)
    want = torch.from_numpy(g["hidden"])
    assert float((h - want).abs().max() / want.abs().max()) < 1e-5 and np.array_equal(am.numpy(), g["mask"])
    # ---- VITS phoneme encoder
    monkeypatch.setattr(pph, "ops", ops_t)
    monkeypatch.setattr(pph, "_DEV", cpu)
    with open(os.path.join(GOLD, "phoneme_keys.json")) as f:
        pshapes = {k: tuple(v) for k, v in json.load(f).items()}
    pe = pph.PhonemeEncoder(**cases.PHONEME)
    pe.load_state_dict(cases.phoneme_state_dict(pshapes), strict=True)
    gp = np.load(os.path.join(GOLD, "phoneme_speech_b4.npz"))
    emb, pm = pe(cases.phoneme_input())
    wantp = torch.from_numpy(gp["emb"])
    assert float((emb - wantp).abs().max() / wantp.abs().max()) < 1e-5 and np.array_equal(pm.numpy(), gp["mask"])


def test_retarget_config_maps_the_reference_configs_onto_the_hip_targets(tmp_path):
    """pipeline.retarget_config / build_model(config=<yaml path>) (pipeline.py:155-157): the reference's own config dicts
    (utils.py:116-561), as dicts and through a YAML file, come out with our targets and untouched params — and equal the
    configs `default_audioldm_config(..., conditioners="hip")` builds."""
    import yaml
    from oracle import refimport
    if not refimport.available():
        pytest.skip("reference not present")
    refimport.install()
    from audioldm2.utils import default_audioldm_config as ref_config
    from audioldm2_amd import pipeline as P

    def targets(node, acc):
        if isinstance(node, dict):
            if "target" in node:
                acc.append(node["target"])
            for v in node.values():
                targets(v, acc)
        return acc

    for name in ("audioldm2-full", "audioldm2-full-large-1150k", "audioldm2-speech-gigaspeech", "audioldm_48k"):
        ref = ref_config(name)
        ours = P.retarget_config(ref)
        assert all(t.startswith("audioldm2_amd.") for t in targets(ours, [])), targets(ours, [])
        assert ref["model"]["target"].startswith("audioldm2.")                       # the input is not modified
        mp = ours["model"]["params"]
        hip = P.default_audioldm_config(name, conditioners="hip")["model"]["params"]
        assert mp["unet_config"] == hip["unet_config"]
        assert mp["first_stage_config"]["params"]["ddconfig"] == hip["first_stage_config"]["params"]["ddconfig"]
        assert "lossconfig" not in mp["first_stage_config"]["params"]                 # training only: no counterpart
        want = P.hip_cond_stage_config(name)
        if "-speech-" in name:   # our config names the device explicitly; the reference's speech config leaves the default
            want["crossattn_audiomae_generated"]["params"].pop("device")
        assert mp["cond_stage_config"] == want
        assert list(mp["cond_stage_config"].keys()) == list(ref["model"]["params"]["cond_stage_config"].keys())
        path = tmp_path / f"{name}.yaml"
        with open(path, "w") as f:
            yaml.safe_dump(ref, f)
        assert P.retarget_config(str(path)) == ours
        assert P.retarget_config(ours) == ours                                        # idempotent


def test_build_model_from_a_reference_yaml_builds_the_hip_module_tree(tmp_path):
    """build_model(config=<path of the reference's YAML>) (pipeline.py:155-157) constructs OUR LatentDiffusion, UNet, VAE and
    conditioner from the reference's config of audioldm_48k (the smallest conditioner stack: CLAP text)."""
    import yaml
    from oracle import refimport
    if not refimport.available():
        pytest.skip("reference not present")
    refimport.install()
    from audioldm2.utils import default_audioldm_config as ref_config
    from audioldm2_amd import pipeline as P
    path = tmp_path / "audioldm_48k.yaml"
    with open(path, "w") as f:
        yaml.safe_dump(ref_config("audioldm_48k"), f)
    ld = P.build_model(config=str(path), model_name="audioldm_48k")
    mods = [type(ld), type(ld.model.diffusion_model), type(ld.first_stage_model)] + [type(m) for m in ld.cond_stage_models]
    assert all(t.__module__.startswith("audioldm2_amd.") for t in mods), mods
    assert ld.sampling_rate == 48000 and ld.first_stage_model.decoder is not None


def test_save_wave_names_files_like_the_reference(tmp_path):
    """pipeline.save_wave vs the reference's utils.save_wave (utils.py:53-77) run with its `soundfile` stubbed: the same paths
    for batches, single clips, list names and names that carry `.wav`; the files are readable 16-bit PCM of the input."""
    from scipy.io import wavfile
    from oracle import refimport
    from audioldm2_amd.pipeline import save_wave
    rng = np.random.default_rng(0)
    cases_ = [(rng.uniform(-0.5, 0.5, (3, 1, 1600)).astype(np.float32), "outwav"),
              (rng.uniform(-0.5, 0.5, (1, 1, 800)).astype(np.float32), "a dog barking"),
              (rng.uniform(-0.5, 0.5, (2, 1, 800)).astype(np.float32), ["x/first.wav", "second"]),
              (rng.uniform(-0.5, 0.5, (1, 1, 800)).astype(np.float32), "y" * 300)]
    ref_paths = None
    if refimport.available():
        refimport.install()
        import audioldm2.utils as ru
        written = []
        ru.sf.write = lambda path, data, samplerate: written.append(path)
        for w, n in cases_:
            ru.save_wave(w, str(tmp_path), name=n, samplerate=16000)
        ref_paths = written
    ours = []
    for w, n in cases_:
        ours += save_wave(w, str(tmp_path), name=n, samplerate=16000)
    if ref_paths is not None:
        assert ours == ref_paths
    sr, data = wavfile.read(ours[0])
    assert sr == 16000 and data.dtype == np.int16 and data.shape == (1600,)
    assert np.abs(data / 32767.0 - cases_[0][0][0, 0]).max() <= 0.5 / 32767 + 1e-9   # libsndfile's float -> PCM16 scale is 0x7FFF


def test_phoneme_ids_and_batch_layout_match_the_reference():
    """phoneme.phoneme_ids vs latent_diffusion/util.py:28-49 (known, unknown, over-long strings); make_batch_for_text_to_audio
    vs pipeline.py:82-121 for a prompt without transcription (every tensor the reference puts into the batch), and the
    transcription path: refused without a phonemizer hook, mapped through it when set."""
    from oracle import refimport
    from audioldm2_amd import pipeline as P
    from audioldm2_amd.phoneme import VITS_SYMBOLS, phoneme_ids
    assert len(VITS_SYMBOLS) == 183
    if refimport.available():
        refimport.install()
        from audioldm2.latent_diffusion.util import get_vits_phoneme_ids_no_padding as ref_ids
        from audioldm2.pipeline import make_batch_for_text_to_audio as ref_batch
        for s in ("", "həlˈoʊ wˈɜːld, ðɪs ɪz ɐ tˈɛst!", "it's 'quoted' § unknown", "a" * 400):
            assert torch.equal(phoneme_ids(s, 3), ref_ids([s] * 3)["phoneme_idx"])
        rb, ob = ref_batch("a dog barking", batchsize=2), P.make_batch_for_text_to_audio("a dog barking", batchsize=2)
        for k, v in rb.items():
            assert k in ob, k
            assert (torch.equal(v, ob[k]) if torch.is_tensor(v) else v == ob[k]), k
    with pytest.raises(RuntimeError, match="TEXT2PHONEME"):
        P.make_batch_for_text_to_audio("x", transcription="hello world")
    P.TEXT2PHONEME = lambda t: "həlˈoʊ"
    try:
        b = P.make_batch_for_text_to_audio("x", transcription="hello", batchsize=2)
        assert torch.equal(b["phoneme_idx"], phoneme_ids("həlˈoʊ", 2))
    finally:
        P.TEXT2PHONEME = None


def test_save_waveform_names_and_normalises_like_the_reference(tmp_path):
    """LatentDiffusion.save_waveform (ddpm.py:1393-1415): `<global_step>_<i>_<name>.wav` for one name, `<name[i]>.wav` (stem of a
    name that carries .wav) for a list; every clip peak-normalised to 0.8."""
    from types import SimpleNamespace
    from scipy.io import wavfile
    from audioldm2_amd.pipeline import LatentDiffusion
    me = SimpleNamespace(global_step=0, sampling_rate=16000)
    w = np.random.default_rng(1).uniform(-0.3, 0.3, (2, 1, 1600)).astype(np.float32)
    p = LatentDiffusion.save_waveform(me, w, str(tmp_path), "outwav")
    assert [os.path.basename(x) for x in p] == ["0_0_outwav.wav", "0_1_outwav.wav"]
    q = LatentDiffusion.save_waveform(me, w, str(tmp_path), ["a/b/first.wav", "second"])
    assert [os.path.basename(x) for x in q] == ["first.wav", "second.wav"]
    sr, d = wavfile.read(p[1])
    assert sr == 16000 and abs(np.abs(d).max() / 32768.0 - 0.8) < 1e-4
    with pytest.raises(NotImplementedError):
        LatentDiffusion.save_waveform(me, w, str(tmp_path), 3)


def test_reranker_without_clap_weights_is_flagged_and_warned():
    """a checkpoint without `clap.*` entries, loaded through the PARENT with strict=False, must leave the
    re-ranker marked as never loaded (PyTorch runs load_state_dict post-hooks on every submodule whether or not a key matched), and
    `_check_candidates` must warn before ranking with a randomly initialised model."""
    import warnings
    from types import SimpleNamespace
    from audioldm2_amd import clap
    from audioldm2_amd.pipeline import LatentDiffusion

    class Parent(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.hot = torch.nn.Linear(4, 4)
            self.clap = clap.CLAPAudioEmbeddingClassifierFreev2(embed_mode="audio", unconditional_prob=0.0, sampling_rate=16000,
                                                                config=cases.clap_text_test_config(),
                                                                audio_config=cases.htsat_test_config())
    m = Parent()
    assert m.clap.weights_loaded is False
    full = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict({k: v for k, v in full.items() if not k.startswith("clap.")}, strict=False)
    assert m.clap.weights_loaded is False, "a load that supplied no clap.* tensor marked the re-ranker as loaded"
    me = SimpleNamespace(clap=m.clap)
    with pytest.warns(UserWarning, match="randomly initialised"):
        LatentDiffusion._check_candidates(me, 2, {"input_ids": None})
    m.load_state_dict(full, strict=False)
    assert m.clap.weights_loaded is True
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        LatentDiffusion._check_candidates(SimpleNamespace(clap=m.clap), 2, {"input_ids": None})
    # a direct load of the module itself counts too, a partial one does not
    own = {k: v.clone() for k, v in m.clap.state_dict().items()}
    fresh = Parent().clap
    fresh.load_state_dict(dict(list(own.items())[:-1]), strict=False)
    assert fresh.weights_loaded is False
    fresh.load_state_dict(own)
    assert fresh.weights_loaded is True


@pytest.mark.timeout(900)
def test_parity_on_checkpoint_script_reference_stage_and_checkpoint_loading(tmp_path):
    """tools/parity_on_checkpoint.py on a random-init checkpoint written in the REFERENCE's format
    (`torch.save({"state_dict": LatentDiffusion.state_dict()})`, pipeline.py:166-177, with the extra `model_ema.*` / `clap.*`
    entries a released checkpoint carries): the reference stage runs the real `LatentDiffusion.generate_batch` from that file and
    writes the cache; `build_model(ckpt_path=...)` — the path the hip stage takes — loads the same hot-path tensors bit for bit and
    refuses a checkpoint with a hot-path tensor missing; without a GPU the hip stage refuses to run (exit code 2, no CPU path)."""
    import subprocess
    import sys
    from oracle import refimport, weights
    if not refimport.available():
        pytest.skip("reference checkout not present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    refimport.install()
    import audioldm2.utils as ru
    from audioldm2.latent_diffusion.models.ddpm import LatentDiffusion as RefLD
    from audioldm2_amd.pipeline import build_model, default_audioldm_config
    P = ru.default_audioldm_config("audioldm2-full")["model"]["params"]
    cond = default_audioldm_config("audioldm2-full")["model"]["params"]["cond_stage_config"]
    for k in cond:
        cond[k]["params"]["device"] = "cpu"
    P["cond_stage_config"], P["device"] = cond, "cpu"
    torch.manual_seed(0)
    sd = RefLD(**P).state_dict()
    hot = {k: tuple(v.shape) for k, v in sd.items() if k.startswith(("model.diffusion_model.", "first_stage_model."))}
    sd.update(weights.make_state_dict(hot, seed=3))
    sd["scale_factor"] = torch.tensor(cases.SCALE_FACTOR)
    sd["model_ema.decay"] = torch.tensor(0.9999)            # entries a released checkpoint has and the hot path ignores
    sd["clap.model.logit_scale_a"] = torch.tensor(1.0)
    ckpt = tmp_path / "random_init_reference_format.pth"
    torch.save({"state_dict": sd, "global_step": 1}, ckpt)
    cache = tmp_path / "ref.npz"
    script = os.path.join(root, "tools", "parity_on_checkpoint.py")
    r = subprocess.run([sys.executable, script, "--ckpt", str(ckpt), "--model", "audioldm2-full", "--steps", "1", "--batch", "1",
                        "--stage", "reference", "--cache", str(cache)], capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    g = np.load(cache)
    assert g["s1_b1_latent"].shape == (1, 8, 256, 16) and g["s1_b1_mel"].shape == (1, 1, 1024, 64)
    assert g["s1_b1_wave"].shape == (1, 1, 163872) and np.isfinite(g["s1_b1_wave"]).all() and float(np.abs(g["s1_b1_wave"]).max()) > 0
    # the hip stage's loader: same tensors, strict on the hot path
    ld = build_model(ckpt_path=str(ckpt), model_name="audioldm2-full")
    mine = ld.state_dict()
    assert "scale_factor" in mine and "alphas_cumprod" in mine
    for k in list(hot)[::97] + [k for k in ("scale_factor", "alphas_cumprod", "betas", "logvar") if k in mine]:
        assert torch.equal(mine[k].cpu().float(), sd[k].float()), k
    bad = dict(sd)
    del bad[next(iter(hot))]
    torch.save({"state_dict": bad}, tmp_path / "bad.pth")
    with pytest.raises(RuntimeError, match="hot-path"):
        build_model(ckpt_path=str(tmp_path / "bad.pth"), model_name="audioldm2-full")
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, script, "--ckpt", str(ckpt), "--stage", "hip", "--cache", str(cache), "--steps", "1",
                            "--batch", "1"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and "needs the MI355X" in r.stderr


def test_shared_cfg_prefix_ends_at_the_first_transformer_with_a_context():
    """UNetModel._shared_prefix_end: the two halves of a classifier-free-guidance batch compute the same values until the first
    SpatialTransformer that receives a context (input block 4, layer 2 in every cross-attention config: ResBlock, context-free
    transformer | AudioMAE transformer ...); the FiLM-conditioned 48 kHz model shares nothing (y enters every ResBlock)."""
    from audioldm2_amd.pipeline import build_model
    from audioldm2_amd.unet import ResBlock, SpatialTransformer
    for name, nctx in (("audioldm2-full", 2), ("audioldm2-full-large-1150k", 2), ("audioldm2-speech-gigaspeech", 1)):
        u = build_model(model_name=name).model.diffusion_model
        bi, li = u._shared_prefix_end([torch.zeros(1)] * nctx)
        assert (bi, li) == (4, 2)
        blk = list(u.input_blocks[bi])
        assert isinstance(blk[0], ResBlock) and isinstance(blk[1], SpatialTransformer) and isinstance(blk[2], SpatialTransformer)
        assert not any(isinstance(l, SpatialTransformer) for b in list(u.input_blocks)[:bi] for l in b)
        assert u._shared_prefix_end([]) is None and u._shared_prefix_end([None, None]) is None
    assert build_model(model_name="audioldm_48k").model.diffusion_model._shared_prefix_end([]) is None


def test_committed_traffic_records_describe_the_igemm_sources_in_the_tree():
    """profiles/r06_pmc_traffic_*.json (HBM bytes per launch of the igemm kernels, which bench.py quotes as roofline.traffic) are
    stamped with the hash of the igemm sources they were collected on; bench.py drops them as stale when neither that nor the
    all-sources hash matches.  The committed records must describe the committed igemm sources."""
    import json
    import os
    from audioldm2_amd.lib import source_hash
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert source_hash("igemm") != source_hash()
    for mode in ("bf16x6", "f16x3", "bf16x3"):
        with open(os.path.join(root, "profiles", f"r06_pmc_traffic_{mode}.json")) as f:
            tj = json.load(f)
        assert tj["igemm_source_hash"] == source_hash("igemm"), mode
        assert tj["kernels"], mode


def test_os_epilogue_staging_is_a_conflict_free_transpose():
    """csrc/igemm_dma_os.h, plain / q / k epilogue (round 6): lane (lc = l & 15, lg = l >> 4) holds rows 4 lg + i (i = 0..3) of column lc of a
    16x16 accumulator tile and stages them at float index (4 i + lg) * 16 + lc; lane l then reads the float4 at position l >> 2, columns
    4 (l & 3)..+3, and stores it as row er = 4 ((l >> 2) & 3) + (l >> 4).  Checked here on the index arithmetic alone: every (row, col) is
    written once, a reader gets the row it stores, and the 64 lanes of one ds_write_b32 hit 64 different 4-byte banks (row-major staging
    put the four k-groups on the same 16 banks).  Same for the GEGLU form's 32 x 8 tile (lanes lc < 8 write, lane l reads position l >> 1)."""
    tile = {}
    for i in range(4):
        banks = set()
        for l in range(64):
            lc, lg = l & 15, l >> 4
            idx = (i * 4 + lg) * 16 + lc
            assert idx not in tile
            tile[idx] = (4 * lg + i, lc)
            banks.add(idx % 64)
        assert len(banks) == 64, i
        assert len({((lg * 4 + i) * 16 + lc) % 64 for lg in range(4) for lc in range(16)}) == 16   # the layout it replaces
    for l in range(64):
        er, ec = ((l >> 2) & 3) * 4 + (l >> 4), (l & 3) * 4
        for c in range(4):
            assert tile[(l >> 2) * 16 + ec + c] == (er, ec + c)
    assert sorted({((l >> 2) & 3) * 4 + (l >> 4) for l in range(64)}) == list(range(16))
    g = {}
    for i in range(4):
        for half in range(2):
            banks = set()
            for l in range(64):
                lc, lg = l & 15, l >> 4
                if lc < 8:
                    idx = (half * 16 + i * 4 + lg) * 8 + lc
                    assert idx not in g
                    g[idx] = (half * 16 + 4 * lg + i, lc)
                    banks.add(idx % 64)
            assert len(banks) == 32
    for l in range(64):
        pos, ec = l >> 1, (l & 1) * 4
        er = (pos & 16) + (pos & 3) * 4 + ((pos & 15) >> 2)
        for c in range(4):
            assert g[pos * 8 + ec + c] == (er, ec + c)


def test_halo_zero_slot_keeps_the_bank_group_of_the_slot_it_replaces():
    """csrc/igemm_dma_halo.h: a lane whose tap falls into the left / right zero padding reads the zero region at byte offset
    (off & 0xF0) instead of the patch slot at `off` — the same 16-byte bank group (bits 4..7 of the LDS byte address), so the 16 lanes of a
    ds_read_b128 phase keep touching 16 different groups; the k-step XOR (1 << 5) and the part offset (q KB) stay inside the NP KB zero
    region."""
    for NP in (2, 3):
        for pp in range(0, 400):
            for lh in (0, 1):
                off = (pp >> 4) * (NP * 1024) + (pp & 15) * 64 + ((((pp >> 2) & 3) ^ lh) << 4)
                z = off & 0xF0
                for step in (0, 1):
                    assert ((z ^ (step << 5)) >> 4) & 15 == ((off ^ (step << 5)) >> 4) & 15
                    for q in range(NP):
                        assert 0 <= (z ^ (step << 5)) + q * 1024 < NP * 1024


def test_committed_traffic_records_carry_the_shipped_geometry_tables_stamp():
    """An instantiation's AVERAGE bytes per launch follows the table that routes geometries to it: the round-6 traffic records are
    stamped with lib.tuning_hash() next to the source hashes, and bench.py reports `same_geometry_tables` from it.  The committed
    records must have been collected under the committed tables."""
    import json
    import os
    from audioldm2_amd.lib import tuning_hash
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = tuning_hash()
    assert len(h) == 16 and h == tuning_hash()
    for mode in ("bf16x6", "f16x3", "bf16x3"):
        with open(os.path.join(root, "profiles", f"r06_pmc_traffic_{mode}.json")) as f:
            assert json.load(f).get("tuning_hash") == h, mode
