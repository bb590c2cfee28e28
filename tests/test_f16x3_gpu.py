"""The "f16x3" operand format (round 6; include/aldm_hip.h aldm_igemm_desc.a_fmt) through the C ABI, against PyTorch on the CPU
evaluated in fp64: GEMMs whose A operand comes out of a GroupNorm or LayerNorm take 2-part IEEE-fp16 images of power-of-two
scaled operands and run hi*hi + hi*lo + lo*hi on the fp16 matrix instruction — three MFMAs per fp32 product instead of six.  It is
held to the SAME fp32-grade bars as the default mode (tests/tolerances.py: one contraction 2e-6, behind a fused normalisation /
activation 5e-6), on normal and on stress inputs (heavy-tailed weights, huge / tiny normalisation gains, channels with |mean| >>
std), on every kernel family that has an fp16 instantiation (classic, loader-wave, halo-patch, operand-stationary), and the image
itself is checked against its definition (22 of 24 significand bits; an absolute floor of 2^-25 / scale under fp16's normal range)."""
import math

import pytest
import torch
from tolerances import F64 as F
from tolerances import fused_tol, log_err

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return log_err(float((a - b).abs().max() / (b.abs().max() + 1e-30)), 0.0)


def cl(x):
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def uncl(y):
    return y.cpu().permute(0, 3, 1, 2).contiguous()


def g(seed=0):
    return torch.Generator().manual_seed(seed)


@pytest.fixture(scope="module")
def ops():
    from audioldm2_amd import ops as o
    prev = o.set_mma("f16x3")
    yield o
    o.set_mma(prev)


def _t3(n, gen):
    """Student-t, 3 degrees of freedom (the heavy-tailed weights of the stress tests)."""
    z = torch.randn(n, generator=gen)
    c = torch.randn(3, *z.shape, generator=gen).pow(2).sum(0) / 3.0
    return z / c.sqrt()


def test_images_are_fp16_pairs_of_scaled_values(ops):
    """gn_split / layernorm(split_out) write fp16 images in this mode: hi + lo reproduces scale * value to 2^-22 relative (plus
    2^-25 absolute in scaled units: lo below fp16's normal range), the scale is a power of two that keeps the a-priori bound of the
    normalisation under 32768, and a split image of RAW values (no bound) stays an exact 3-part bf16 image."""
    B, C, H, W = 2, 128, 16, 8
    x = torch.randn(B, C, H, W, generator=g(1)) * 3 + 1
    ga, be = torch.randn(C, generator=g(2)) * 1.5, torch.randn(C, generator=g(3))
    s, raw = ops.gn_split(cl(x), ga.cuda(), be.cuda(), groups=32, eps=1e-5, act=ops.ACT_SILU, want_raw=True)
    assert s.fmt == "f16" and s.parts == 2 and raw.fmt == "bf16" and raw.parts == 3
    assert math.log2(s.scale) == int(math.log2(s.scale))
    n = (C // 32) * H * W
    bound = math.sqrt(n) * float(ga.abs().max()) + float(be.abs().max())
    assert 16384.0 < s.scale * bound <= 32768.0
    ref = F.silu(F.group_norm(x, 32, ga, be, 1e-5)).permute(0, 2, 3, 1)
    got = s.float().double().cpu()
    err = (got - ref).abs()
    # (the normalisation itself is fp32: allow its rounding next to the image's)
    assert bool((err <= ref.abs() * 2.0 ** -20 + 2.0 ** -24 / s.scale + 1e-6 * ref.abs().max()).all())
    assert torch.equal(raw.float().cpu(), cl(x).cpu())
    y, so = ops.layernorm(cl(x).view(B, H * W, C), ga.cuda(), be.cuda(), 1e-5, split_out="also")
    assert so.fmt == "f16"
    bound = math.sqrt(C) * float(ga.abs().max()) + float(be.abs().max())
    assert 16384.0 < so.scale * bound <= 32768.0
    d = (so.float().double().cpu() - y.double().cpu()).abs()
    assert bool((d <= y.double().cpu().abs() * 2.0 ** -21 + 2.0 ** -24 / so.scale).all())


_CONV_FORMS = [None, (256, 128, 402), (128, 128, 403), (128, 128, 413), (128, 128, 4), (128, 128, 204), (64, 128, 4), (128, 64, 203),
               (64, 64, 3)]


@pytest.mark.parametrize("form", _CONV_FORMS)
def test_groupnorm_silu_conv3x3_on_every_kernel_family(ops, form):
    """GroupNorm -> SiLU -> conv3x3 (+ bias, timestep row bias, residual): the ResBlock launch (openaimodel.py:280-300), its A operand
    an fp16 image, on the tuned choice and forced onto the halo-patch, classic and loader-wave kernels."""
    B, C, N, H, W = 2, 128, 192, 32, 16
    x = torch.randn(B, C, H, W, generator=g(1)) * 2 + 0.5
    ga, be = torch.randn(C, generator=g(2)) + 1.0, torch.randn(C, generator=g(3)) * 0.3
    w = torch.randn(N, C, 3, 3, generator=g(4)) / math.sqrt(C * 9)
    b = torch.randn(N, generator=g(5))
    emb = torch.randn(B, N, generator=g(6))
    res = torch.randn(B, N, H, W, generator=g(7))
    ref = F.conv2d(F.silu(F.group_norm(x, 32, ga, be, 1e-5)), w, b, padding=1) + emb[:, :, None, None] + res
    pw = ops.pack_conv(w, b)
    a = ops.gn_split(cl(x), ga.cuda(), be.cuda(), groups=32, eps=1e-5, act=ops.ACT_SILU)
    assert a.fmt == "f16"
    if form:
        ops.igemm_force(form[0], form[1], 1, 0, form[2])
    try:
        y, so = ops.conv(a, pw, pad=(1, 1), rowbias=emb.cuda(), res=cl(res), split_out="also")
    finally:
        ops.igemm_force(0, 0, 0)
    assert so.fmt == "bf16" and so.parts == 3 and torch.equal(so.float(), y)   # the epilogue's image: exact 3-part bf16
    assert rel_err(uncl(y), ref) < fused_tol("f16x3")


@pytest.mark.parametrize("K,N,M,form", [(256, 256, 4096, None), (256, 256, 4096, (32, 128, 302)), (256, 256, 4096, (32, 128, 304)),
                                        (384, 384, 1024, (32, 128, 303)), (256, 768, 2048, (64, 128, 4)), (640, 640, 1024, (64, 64, 203)),
                                        (1024, 256, 4096, (128, 128, 204)), (2048, 640, 512, None)])
def test_layernorm_linear(ops, K, N, M, form):
    """LayerNorm -> Linear (+ bias + residual): to_q / proj launches; the operand-stationary plain form, classic and loader-wave tiles."""
    x = torch.randn(1, M, K, generator=g(1)) * 1.7 - 0.4
    ga, be = torch.randn(K, generator=g(2)) * 0.5 + 1.0, torch.randn(K, generator=g(3)) * 0.2
    w = torch.randn(N, K, generator=g(4)) / math.sqrt(K)
    b = torch.randn(N, generator=g(5))
    res = torch.randn(1, M, N, generator=g(6))
    ref = F.layer_norm(x, (K,), ga, be, 1e-5) @ w.double().t() + b.double() + res.double()
    pw = ops.pack_conv(w, b)
    n = ops.layernorm(x.cuda(), ga.cuda(), be.cuda(), 1e-5, split_out="only")
    if form:
        ops.igemm_force(form[0], form[1], 1, 0, form[2])
    try:
        y = ops.linear(n, pw, res=res.cuda())
    finally:
        ops.igemm_force(0, 0, 0)
    assert rel_err(y, ref) < fused_tol("f16x3")


@pytest.mark.parametrize("C,M,form", [(256, 4096, None), (256, 4096, (32, 128, 303)), (384, 1024, (32, 128, 302)), (640, 256, None),
                                      (640, 1024, (256, 128, 2))])
def test_layernorm_geglu_ff_out(ops, C, M, form):
    """LayerNorm -> GEGLU projection -> FF-out (attention.py:37-63), all three-product: the GEGLU output has an a-priori bound as
    well — |value * gelu(gate)| <= |value| |gate| <= (R c + b)^2, R the LayerNorm rows' 2-norm bound, c the weight's largest column
    norm — so the epilogue writes it as an fp16 image (operand-stationary and classic epilogues) and the FF-out GEMM reads that."""
    x = torch.randn(1, M, C, generator=g(1))
    ga, be = torch.randn(C, generator=g(2)) * 0.3 + 1.0, torch.randn(C, generator=g(3)) * 0.1
    w = torch.randn(8 * C, C, generator=g(4)) / math.sqrt(C)
    b = torch.randn(8 * C, generator=g(5)) * 0.5
    w2 = torch.randn(C, 4 * C, generator=g(6)) / math.sqrt(4 * C)
    b2 = torch.randn(C, generator=g(7))
    h = F.layer_norm(x, (C,), ga, be, 1e-5) @ w.double().t() + b.double()
    ref = h[..., :4 * C] * F.gelu(h[..., 4 * C:])
    ref2 = ref @ w2.double().t() + b2.double() + x.double()
    pw, pw2 = ops.pack_geglu(w, b), ops.pack_conv(w2, b2)
    n = ops.layernorm(x.cuda(), ga.cuda(), be.cuda(), 1e-5, split_out="only")
    assert n.rn > 0.0
    if form:
        ops.igemm_force(form[0], form[1], 1, 0, form[2])
    try:
        y, so = ops.linear_geglu(n, pw, split_out="also")
    finally:
        ops.igemm_force(0, 0, 0)
    assert rel_err(y, ref) < fused_tol("f16x3")
    assert so.fmt == "f16" and so.parts == 2
    assert float(so.scale) * float(ref.abs().max()) <= 32768.0          # the bound held
    d = (so.float().double().cpu() - y.double().cpu()).abs()
    assert bool((d <= y.double().cpu().abs() * 2.0 ** -21 + 2.0 ** -24 / so.scale).all())
    y2 = ops.linear(so, pw2, res=x.cuda())
    assert rel_err(y2, ref2) < fused_tol("f16x3")
    # and with the switch off the image is the exact 3-part bf16 one
    ops.F16_FF_OUT = False
    try:
        y3, so3 = ops.linear_geglu(n, pw, split_out="also")
    finally:
        ops.F16_FF_OUT = True
    assert so3.fmt == "bf16" and so3.parts == 3 and torch.equal(so3.float(), y3)


@pytest.mark.parametrize("B,L,heads,form", [(2, 256, 8, None), (2, 256, 8, (32, 128, 303)), (2, 128, 12, (32, 128, 302)), (1, 64, 20, None),
                                            (16, 1024, 8, None), (2, 96, 2, (64, 64, 3)), (1, 32, 2, (64, 64, 3))])
def test_layernorm_qkv_attention(ops, B, L, heads, form):
    """LayerNorm -> fused q | k | v projection -> self-attention, three-product throughout: a LayerNorm-fed projection is bounded by
    R c before the data exists, so the QKV epilogue (operand-stationary and classic) writes K and V^T as fp16 images of power-of-two
    scaled values, the attention kernel splits q and the probabilities (x 2^15) into fp16 parts and runs both contractions on the
    fp16 matrix instruction, with an integer softmax reference (exact offsets).  Checked against fp64 at the fused fp32-grade bar,
    at 1, 2, 3 and many key tiles, 32 and 64 queries per wave; with ALDM_F16_ATTN off the images are 3-part bf16 and the attention
    runs the six-product kernel behind the three-product projection."""
    C = heads * 32
    x = torch.randn(B, L, C, generator=g(1))
    ga, be = torch.randn(C, generator=g(2)) * 0.3 + 1.0, torch.randn(C, generator=g(3)) * 0.1
    wq, wk, wv = (torch.randn(C, C, generator=g(4 + i)) / math.sqrt(C) for i in range(3))
    pw = ops.pack_conv(torch.cat([wq, wk, wv], 0))
    n = ops.layernorm(x.cuda(), ga.cuda(), be.cuda(), 1e-5, split_out="only")
    xn = F.layer_norm(x, (C,), ga, be, 1e-5)
    sh = lambda t: t.view(B, L, heads, 32).transpose(1, 2)
    qd, kd, vd = xn @ wq.double().t(), xn @ wk.double().t(), xn @ wv.double().t()
    ref = F.scaled_dot_product_attention(sh(qd), sh(kd), sh(vd)).transpose(1, 2).reshape(B, L, C)
    for f16_attn in (True, False):
        ops.F16_ATTN = f16_attn
        try:
            if form:
                ops.igemm_force(form[0], form[1], 1, 0, form[2])
            try:
                q, kimg, vtimg = ops.linear_qkv(n, pw, heads, L)
            finally:
                ops.igemm_force(0, 0, 0)
            assert kimg.shape[2] == (2 if f16_attn else 3) and vtimg.shape[3] == kimg.shape[2]
            a, so = ops.attention_presplit(q, kimg, vtimg, heads, split_out="also")
        finally:
            ops.F16_ATTN = True
        if f16_attn:
            qs, ks, vs = kimg._aldm_f16
            assert float(kd.abs().max()) * ks <= 32768.0 and float(vd.abs().max()) * vs <= 32768.0      # the bounds held
            assert float(qd.abs().max()) * (32 ** -0.5) * 1.4426950408889634 * qs <= 32768.0
            kf = kimg.view(torch.float16).float()
            kf = ((kf[:, :, 0] + kf[:, :, 1]) / ks).reshape(B, L, C)
            assert rel_err(kf, kd) < fused_tol("f16x3")
        if f16_attn:   # the output image is an fp16 one under V's scale (a convex combination of the values cannot exceed them)
            assert so.fmt == "f16" and so.scale == vs and float(ref.abs().max()) * vs <= 32768.0
            assert rel_err(so.float(), a.double()) < 2.0 ** -21
        else:
            assert so.fmt == "bf16" and so.parts == 3 and torch.equal(so.float(), a)
        assert rel_err(q, qd) < fused_tol("f16x3")
        assert rel_err(a, ref) < fused_tol("f16x3")
        # ... and the to_out projection (+ bias + residual) consumes it: three products behind the fp16 attention, six otherwise
        wo, bo = torch.randn(C, C, generator=g(9)) / math.sqrt(C), torch.randn(C, generator=g(10)) * 0.1
        y = ops.linear(so, ops.pack_conv(wo, bo), res=x.cuda())
        assert rel_err(y, ref @ wo.double().t() + bo.double() + x.double()) < fused_tol("f16x3")


def test_f16_attention_survives_extreme_scores(ops):
    """Scores hundreds of log2 units apart, arriving in later key tiles: the integer softmax reference follows the running maximum
    (rescale factors are exact powers of two), nothing overflows the 2^15-scaled fp16 probabilities, outputs stay finite and
    within the fused bar plus the score-rounding allowance of the six-product kernel's own test."""
    B, L, heads = 2, 128, 2
    C = heads * 32
    x = torch.randn(B, L, C, generator=g(1))
    ga, be = torch.ones(C), torch.zeros(C)
    wq, wk, wv = (torch.randn(C, C, generator=g(4 + i)) / math.sqrt(C) for i in range(3))
    wq = wq * 12.0
    wk = wk * 12.0     # scores ~ N(0, 144 * ...) in natural units: hundreds of log2 units between keys
    pw = ops.pack_conv(torch.cat([wq, wk, wv], 0))
    n = ops.layernorm(x.cuda(), ga.cuda(), be.cuda(), 1e-5, split_out="only")
    q, kimg, vtimg = ops.linear_qkv(n, pw, heads, L)
    assert kimg.shape[2] == 2
    a = ops.attention_presplit(q, kimg, vtimg, heads)
    assert bool(torch.isfinite(a).all())
    xn = F.layer_norm(x, (C,), ga, be, 1e-5)
    sh = lambda t: t.view(B, L, heads, 32).transpose(1, 2)
    qd, kd, vd = sh(xn @ wq.double().t()), sh(xn @ wk.double().t()), sh(xn @ wv.double().t())
    sc = qd @ kd.transpose(-1, -2) / math.sqrt(32.0)
    ref = (torch.softmax(sc, -1) @ vd).transpose(1, 2).reshape(B, L, C)
    smax = float(sc.abs().max())
    assert smax > 100.0
    assert rel_err(a, ref) < fused_tol("f16x3") + 2.0 ** -24 * smax * 1.4426950408889634


@pytest.mark.parametrize("gain,offset,wkind", [(1e-3, 0.0, "normal"), (50.0, 10.0, "normal"), (1.0, 0.0, "t3"), (1.0, 0.0, "tiny"),
                                               (1.0, 0.0, "huge"), (4.0, -3.0, "t3")])
def test_stress_operands(ops, gain, offset, wkind):
    """What the power-of-two scales must absorb: normalisation gains from 1e-3 to 50 with offsets, input channels with |mean| = 10^3
    std, heavy-tailed (Student-t, 3 dof) weights, weights of magnitude 1e-6 and 1e+4 — the scale comes from the bound, never from
    the data, and nothing overflows or loses its low part."""
    B, C, N, H, W = 2, 128, 128, 32, 16
    gen = g(11)
    mu = 1000.0 * torch.randn(1, C, 1, 1, generator=gen)
    x = mu + torch.randn(B, C, H, W, generator=gen)
    ga = gain * (torch.randn(C, generator=gen) * 0.3 + 1.0)
    be = offset + torch.randn(C, generator=gen) * 0.1 * max(abs(offset), 1.0)
    if wkind == "t3":
        w = _t3((N, C, 3, 3), gen) / math.sqrt(C * 9)
    else:
        w = torch.randn(N, C, 3, 3, generator=gen) / math.sqrt(C * 9) * {"normal": 1.0, "tiny": 1e-6, "huge": 1e4}[wkind]
    ref = F.conv2d(F.silu(F.group_norm(x, 32, ga, be, 1e-5)), w, None, padding=1)
    pw = ops.pack_conv(w, None)
    a = ops.gn_split(cl(x), ga.cuda(), be.cuda(), groups=32, eps=1e-5, act=ops.ACT_SILU)
    y = ops.conv(a, pw, pad=(1, 1))
    assert bool(torch.isfinite(y).all())
    # GroupNorm at |mean| / std = 10^3 carries its own fp32 rounding (5e-5 of the normalised value, tests/test_ops_gpu.py): compare
    # with the SAME launch in the default mode, and with fp64 at the GroupNorm stress bar
    prev = ops.set_mma("bf16x6")
    try:
        y6 = ops.conv(ops.gn_split(cl(x), ga.cuda(), be.cuda(), groups=32, eps=1e-5, act=ops.ACT_SILU), pw, pad=(1, 1))
    finally:
        ops.set_mma(prev)
    assert rel_err(y, y6) < fused_tol("f16x3")
    assert rel_err(uncl(y), ref) < 1e-4


def test_out_of_bound_values_saturate_instead_of_overflowing(ops):
    """The seat belt behind the a-priori bounds: a producer handed a scale that is too large for its data (which the bounds rule out:
    here it is forced through the C entry point) writes +-65504, never inf / NaN, and values inside the range are unaffected."""
    from audioldm2_amd import lib as L
    M, C = 64, 128
    x = torch.randn(M, C, generator=g(5))
    ga, be = torch.ones(C), torch.zeros(C)
    ref = F.layer_norm(x, (C,), ga, be, 1e-5).double()
    for scale in (2.0 ** 20, 2.0 ** 15):      # |scale * value| up to ~4e6 (all saturate) / ~1.2e5 (the values beyond +-2 saturate)
        so = ops.SplitT.empty((M, C), torch.device("cuda"), f16_scale=scale)
        xd, gd, bd = x.cuda(), ga.cuda(), be.cuda()     # (held: the entry point takes raw pointers)
        L.check(L.load().aldm_layernorm_split_f16(xd.data_ptr(), None, so.data_ptr(), M, C, gd.data_ptr(), bd.data_ptr(),
                                                  1e-5, scale, torch.cuda.current_stream().cuda_stream), "layernorm_split_f16")
        torch.cuda.synchronize()
        h = so.data.view(torch.float16).float().cpu()
        assert bool(torch.isfinite(h).all())
        got = (h[:, :, 0].double() + h[:, :, 1].double()).reshape(M, C)
        want = (ref * scale).clamp(-65504.0, 65504.0)
        assert float((got - want).abs().max()) <= 65504.0 * 2.0 ** -20
        assert float(got.abs().max()) == 65504.0          # something did saturate in both cases


def test_mixed_graph_of_a_resblock_and_a_transformer_block_matches_the_default_mode(ops):
    """A UNet in small (tests' tiny config) evaluated in f16x3 and in bf16x6 on the same weights and inputs: same result to the
    fp32-grade UNet bar — the mode changes which matrix instruction runs, not what is computed."""
    from audioldm2_amd.unet import UNetModel
    from oracle import cases, weights
    cfg = cases.UNET_TINY
    unet = UNetModel(**cfg)
    unet.load_state_dict(weights.make_state_dict(weights.shapes_of(unet), seed=0))
    x, t, ctxs, masks, _ = cases.unet_inputs(cfg, 2, 16, 8)
    kw = dict(context_list=[c.cuda() for c in ctxs], context_attn_mask_list=[m.cuda() for m in masks])
    e16 = unet(x.cuda(), t.cuda(), **kw).clone()
    prev = ops.set_mma("bf16x6")
    try:
        unet.drop_step_caches()
        e6 = unet(x.cuda(), t.cuda(), **kw).clone()
    finally:
        ops.set_mma(prev)
        unet.drop_step_caches()
    assert rel_err(e16, e6) < 5e-6


@pytest.mark.parametrize("seed,span", [(1, 1.5), (2, 1.5), (3, 2.5)])
def test_unet_with_per_tensor_gains_matches_the_default_mode(ops, seed, span):
    """Random-init networks are tame: every layer has the same magnitude.  A trained checkpoint does not, and the a-priori scales of
    this mode are functions of the parameters (max|gamma|, ||beta||, the weights' column norms).  Every floating parameter tensor of
    the tiny UNet gets its own gain 2^u, u uniform in [-span, span] (norm gains and biases included: bounds from 0.18x to 5.7x their
    random-init size, attention logits up to 30x): the mode must still agree with bf16x6 at the fp32-grade UNet bar, with finite output."""
    from audioldm2_amd.unet import UNetModel
    from oracle import cases, weights
    cfg = cases.UNET_TINY
    unet = UNetModel(**cfg)
    sd = weights.make_state_dict(weights.shapes_of(unet), seed=0)
    gen = g(100 + seed)
    for k, v in sd.items():
        if torch.is_floating_point(v):
            sd[k] = v * float(2.0 ** ((torch.rand(1, generator=gen).item() * 2.0 - 1.0) * span))
    unet.load_state_dict(sd)
    x, t, ctxs, masks, _ = cases.unet_inputs(cfg, 2, 16, 8)
    kw = dict(context_list=[c.cuda() for c in ctxs], context_attn_mask_list=[m.cuda() for m in masks])
    e16 = unet(x.cuda(), t.cuda(), **kw).clone()
    assert bool(torch.isfinite(e16).all())
    prev = ops.set_mma("bf16x6")
    try:
        unet.drop_step_caches()
        e6 = unet(x.cuda(), t.cuda(), **kw).clone()
    finally:
        ops.set_mma(prev)
        unet.drop_step_caches()
    assert rel_err(e16, e6) < 1e-5
