"""Per-op parity of the HIP kernels (through the C ABI) against PyTorch on the CPU evaluated in fp64 on the same fp32 inputs.

Tolerances (tests/tolerances.py, per product mode, <= 5x the measured error): a contraction in an fp32-grade mode (fp32 MFMA,
bf16x6) max|err| / max|ref| <= 2e-6, behind a fused GroupNorm / activation prologue or in front of a GELU / softmax <= 5e-6;
bf16x3 (16-bit operand significands) <= 5e-5; pure elementwise ops <= 2e-6.
"""
import math

import pytest
import torch
from tolerances import F64 as F   # references in fp64 (every floating argument promoted)
from tolerances import fused_tol, gemm_tol, log_err, stress_tol

pytestmark = pytest.mark.gpu

EW_TOL = 2e-6


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return log_err(float((a - b).abs().max() / (b.abs().max() + 1e-30)), 0.0)


def cl(x):  # NCHW -> NHWC on GPU
    return x.permute(0, 2, 3, 1).contiguous().cuda()


def uncl(y):  # NHWC (GPU) -> NCHW CPU
    return y.cpu().permute(0, 3, 1, 2).contiguous()


@pytest.fixture(scope="module", params=["f32", "bf16x6"])
def ops(request):
    """Every op test runs on both matrix-core paths of the igemm engine: the fp32 MFMA and the bf16-split
    ("BF16x6": fp32 = 6 bf16 partial products of exact operand splits, fp32 accumulate) kernels — same
    tolerances, the split path is fp32-grade."""
    from audioldm2_amd import ops as o
    prev = o.set_mma(request.param)
    yield o
    o.set_mma(prev)


def g(seed=0):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("B,C,N,H,W,k,s,p", [
    (2, 128, 128, 32, 16, 3, 1, 1),     # UNet level-0 ResBlock conv
    (1, 128, 128, 300, 16, 3, 1, 1),    # M = 4800 -> 128x128 tiles incl. ragged M
    (2, 256, 256, 16, 8, 3, 2, 1),      # Downsample stride 2
    (3, 8, 128, 20, 16, 3, 1, 1),       # conv_in: K = 72 (ragged K tile)
    (2, 128, 8, 20, 16, 3, 1, 1),       # out conv: N = 8 (padded N tile)
    (2, 640, 640, 4, 2, 3, 1, 1),       # deepest level, tiny M
    (2, 256, 384, 9, 5, 1, 1, 0),       # 1x1 skip conv, odd extents
    (1, 64, 64, 50, 30, 3, 1, 1),       # N = 64 tile
])
def test_conv2d(ops, B, C, N, H, W, k, s, p):
    x = torch.randn(B, C, H, W, generator=g(1))
    w = torch.randn(N, C, k, k, generator=g(2)) / math.sqrt(C * k * k)
    b = torch.randn(N, generator=g(3))
    ref = F.conv2d(x, w, b, stride=s, padding=p)
    pw = ops.pack_conv(w, b)
    y = ops.conv(cl(x), pw, stride=(s, s), pad=(p, p))
    assert rel_err(uncl(y), ref) < gemm_tol()


def test_conv_fused_prologue_epilogue(ops):
    """skip-concat + GroupNorm(+SiLU) prologue + timestep row-bias + residual epilogue
    (= one half of a UNet output ResBlock, openaimodel.py:280-300, :879)."""
    B, C1, C2, N, H, W = 2, 200, 184, 128, 16, 8  # group 16 straddles x1|x2
    x1 = torch.randn(B, C1, H, W, generator=g(1)) * 2 + 0.5
    x2 = torch.randn(B, C2, H, W, generator=g(2))
    w = torch.randn(N, C1 + C2, 3, 3, generator=g(3)) / math.sqrt((C1 + C2) * 9)
    bias = torch.randn(N, generator=g(4))
    gamma = torch.randn(C1 + C2, generator=g(5))
    beta = torch.randn(C1 + C2, generator=g(6))
    emb = torch.randn(B, N, generator=g(7))
    res = torch.randn(B, N, H, W, generator=g(8))
    xc = torch.cat([x1, x2], 1)
    h = F.silu(F.group_norm(xc, 32, gamma, beta, eps=1e-5))
    ref = F.conv2d(h, w, bias, padding=1) + emb[:, :, None, None] + res
    pw = ops.pack_conv(w, bias)
    a, b2 = cl(x1), cl(x2)
    sc, sh = ops.gn_stats(a, gamma.cuda(), beta.cuda(), groups=32, eps=1e-5, x2=b2)
    # GN scale/shift themselves
    mean = xc.view(B, 32, -1).mean(-1)
    var = xc.view(B, 32, -1).var(-1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    sc_ref = rstd.repeat_interleave((C1 + C2) // 32, 1) * gamma
    assert rel_err(sc, sc_ref) < 1e-5
    y = ops.conv(a, pw, x2=b2, pad=(1, 1), pre=(sc, sh), pre_act=ops.ACT_SILU,
                 rowbias=emb.cuda(), res=cl(res))
    assert rel_err(uncl(y), ref) < fused_tol()


@pytest.mark.parametrize("bm,bn", [(128, 128), (128, 64), (64, 128), (64, 64), (128, 32)])
@pytest.mark.parametrize("splits", [1, 3])
def test_igemm_every_tile_and_splitk(ops, bm, bn, splits):
    """Every block-tile instantiation, with and without split-K (partial tiles -> workspace ->
    deterministic reduce + epilogue), on a ragged M / ragged K-split problem with the full epilogue."""
    B, C, N, H, W = 3, 136, 96, 13, 7  # M = 273 (ragged), K = 1224 = 38.25 k-tiles
    x = torch.randn(B, C, H, W, generator=g(1))
    w = torch.randn(N, C, 3, 3, generator=g(2)) / math.sqrt(C * 9)
    b = torch.randn(N, generator=g(3))
    emb = torch.randn(B, 2 * N, generator=g(4))
    res = torch.randn(B, N, H, W, generator=g(5))
    ref = F.silu(F.conv2d(x, w, b, padding=1) + emb[:, N:, None, None]) + res
    pw = ops.pack_conv(w, b)
    ops.igemm_force(bm, bn, splits)
    try:
        y1 = ops.conv(cl(x), pw, pad=(1, 1), rowbias=emb.cuda()[:, N:], act=ops.ACT_SILU, res=cl(res))
        y2 = ops.conv(cl(x), pw, pad=(1, 1), rowbias=emb.cuda()[:, N:], act=ops.ACT_SILU, res=cl(res))
    finally:
        ops.igemm_force(0, 0, 0)
    assert rel_err(uncl(y1), ref) < fused_tol()
    assert torch.equal(y1, y2), "split-K reduce must be bitwise reproducible"


@pytest.mark.parametrize("splits", [1, 3])
@pytest.mark.parametrize("with_pre", [False, True])
def test_igemm_two_wave_groups(ops, splits, with_pre):
    """KGRP = 2: two 4-wave groups share one 64x64 tile's K loop (odd tile count: one group runs a
    barrier-only iteration) and add their accumulators through LDS; alone and combined with split-K."""
    B, C, N, H, W = 3, 136, 96, 13, 7  # K = 1224 -> 39 k-tiles
    x = torch.randn(B, C, H, W, generator=g(1))
    w = torch.randn(N, C, 3, 3, generator=g(2)) / math.sqrt(C * 9)
    b = torch.randn(N, generator=g(3))
    res = torch.randn(B, N, H, W, generator=g(5))
    gamma, beta = torch.randn(C, generator=g(6)), torch.randn(C, generator=g(7))
    a = cl(x)
    kw = {}
    xin = x
    if with_pre:
        sc, sh = ops.gn_stats(a, gamma.cuda(), beta.cuda(), groups=2, eps=1e-5)
        kw = dict(pre=(sc, sh), pre_act=ops.ACT_SILU)
        xin = F.silu(F.group_norm(x, 2, gamma, beta, eps=1e-5))
    ref = F.conv2d(xin, w, b, padding=1) + res
    pw = ops.pack_conv(w, b)
    ops.igemm_force(64, 64, splits, 2)
    try:
        y1 = ops.conv(a, pw, pad=(1, 1), res=cl(res), **kw)
        y2 = ops.conv(a, pw, pad=(1, 1), res=cl(res), **kw)
    finally:
        ops.igemm_force(0, 0, 0, 0)
    assert rel_err(uncl(y1), ref) < fused_tol()
    assert torch.equal(y1, y2)


def test_igemm_auto_splitk_deep_level(ops):
    """The shape class that triggers automatic split-K (deepest UNet level: M = 1024, K = 5760) with
    the GroupNorm+SiLU prologue, against the same launch with split-K disabled."""
    B, C, N, H, W = 16, 640, 640, 32, 2
    x = torch.randn(B, C, H, W, generator=g(1))
    w = torch.randn(N, C, 3, 3, generator=g(2)) / math.sqrt(C * 9)
    gamma, beta = torch.randn(C, generator=g(3)), torch.randn(C, generator=g(4))
    ref = F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-5)), w, None, padding=1)
    a = cl(x)
    sc, sh = ops.gn_stats(a, gamma.cuda(), beta.cuda(), groups=32, eps=1e-5)
    pw = ops.pack_conv(w)
    y = ops.conv(a, pw, pad=(1, 1), pre=(sc, sh), pre_act=ops.ACT_SILU)
    assert rel_err(uncl(y), ref) < fused_tol()
    ops.igemm_force(64, 64, 1)
    try:
        y1 = ops.conv(a, pw, pad=(1, 1), pre=(sc, sh), pre_act=ops.ACT_SILU)
    finally:
        ops.igemm_force(0, 0, 0)
    assert rel_err(y, y1) < 1e-5


@pytest.mark.parametrize("bm,bn", [(128, 128), (64, 128), (128, 64)])
@pytest.mark.parametrize("uniform", [False, True])
@pytest.mark.parametrize("splits", [1, 3])
def test_igemm_eight_wave_tiles(ops, bm, bn, uniform, splits):
    """The 8-wavefront variants of the 128x128 / 64x128 / 128x64 tiles (GroupNorm+SiLU prologue; per-row and
    per-tile-uniform scale/shift loads), alone and with split-K: same results as the 4-wave tile, bit for bit
    (the K order per output element is the same), and within tolerance of the fp32 reference."""
    B, C, N = 3, 136, 96
    H, W = (32, 8) if uniform else (13, 7)  # 256 output positions per sample: every tile inside one sample
    x = torch.randn(B, C, H, W, generator=g(1))
    w = torch.randn(N, C, 3, 3, generator=g(2)) / math.sqrt(C * 9)
    b = torch.randn(N, generator=g(3))
    res = torch.randn(B, N, H, W, generator=g(5))
    gamma, beta = torch.randn(C, generator=g(6)), torch.randn(C, generator=g(7))
    a = cl(x)
    sc, sh = ops.gn_stats(a, gamma.cuda(), beta.cuda(), groups=2, eps=1e-5)
    ref = F.conv2d(F.silu(F.group_norm(x, 2, gamma, beta, eps=1e-5)), w, b, padding=1) + res
    pw = ops.pack_conv(w, b)
    ops.igemm_force(bm, bn, splits)
    try:
        assert ops.igemm_wave8(15) == 15
        y8 = ops.conv(a, pw, pad=(1, 1), pre=(sc, sh), pre_act=ops.ACT_SILU, res=cl(res))
        y8n = ops.conv(a, pw, pad=(1, 1), res=cl(res))  # bit 3: no prologue, 128x128 only
        ops.igemm_wave8(0)
        y4 = ops.conv(a, pw, pad=(1, 1), pre=(sc, sh), pre_act=ops.ACT_SILU, res=cl(res))
        y4n = ops.conv(a, pw, pad=(1, 1), res=cl(res))
    finally:
        ops.igemm_wave8(-1)
        ops.igemm_force(0, 0, 0)
    assert rel_err(uncl(y8), ref) < fused_tol()
    assert torch.equal(y8, y4) and torch.equal(y8n, y4n)


@pytest.mark.parametrize("mode", ["affine", "lrelu", "silu_only", "affine_gelu"])
def test_igemm_prologue_modes(ops, mode):
    """Each templated operand-prologue (GN apply, leaky_relu, and the generic runtime path)."""
    B, C, N, L = 2, 64, 48, 333
    x = torch.randn(B, C, 1, L, generator=g(1))
    w = torch.randn(N, C, 1, 7, generator=g(2)) / math.sqrt(C * 7)
    sc = torch.rand(B, C, generator=g(3)) + 0.5
    sh = torch.randn(B, C, generator=g(4))
    aff = x * sc[:, :, None, None] + sh[:, :, None, None]
    kw = {}
    if mode == "affine":
        h = aff
        kw = dict(pre=(sc.cuda(), sh.cuda()))
    elif mode == "lrelu":
        h = F.leaky_relu(x, 0.1)
        kw = dict(pre_act=ops.ACT_LRELU, pre_slope=0.1)
    elif mode == "silu_only":
        h = F.silu(x)
        kw = dict(pre_act=ops.ACT_SILU)
    else:
        h = F.gelu(aff)
        kw = dict(pre=(sc.cuda(), sh.cuda()), pre_act=ops.ACT_GELU)
    ref = F.conv2d(h, w, None, padding=(0, 9), dilation=(1, 3))
    y = ops.conv(cl(x), ops.pack_conv(w), pad=(0, 9), dil=(1, 3), **kw)
    assert rel_err(uncl(y), ref) < fused_tol()


@pytest.mark.parametrize("M,C,inner", [(1024, 256, 1024), (300, 384, 1536), (64, 640, 2560), (70, 64, 32)])
def test_linear_geglu_fused(ops, M, C, inner):
    """GEGLU (attention.py:37-45) fused into the projection GEMM's epilogue vs proj -> chunk -> x*gelu(gate)."""
    x = torch.randn(M, C, generator=g(1))
    w = torch.randn(2 * inner, C, generator=g(2)) / math.sqrt(C)
    b = torch.randn(2 * inner, generator=g(3))
    val, gate = F.linear(x, w, b).chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    y = ops.linear_geglu(x.cuda(), ops.pack_geglu(w, b))
    assert y.shape == (M, inner)
    assert rel_err(y, ref) < fused_tol()


@pytest.mark.parametrize("bm,bn", [(128, 128), (64, 128)])
def test_linear_geglu_forced_tiles(ops, bm, bn):
    """The GEGLU epilogue on both 128-column tiles (bf16-split path: the 128x128 tile runs 8 waves as 4x2)."""
    M, C, inner = 1000, 256, 512
    x = torch.randn(M, C, generator=g(1))
    w = torch.randn(2 * inner, C, generator=g(2)) / math.sqrt(C)
    b = torch.randn(2 * inner, generator=g(3))
    val, gate = F.linear(x, w, b).chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    pw = ops.pack_geglu(w, b)
    ops.igemm_force(bm, bn, 1)
    try:
        y = ops.linear_geglu(x.cuda(), pw)
    finally:
        ops.igemm_force(0, 0, 0)
    assert rel_err(y, ref) < fused_tol()


def test_weight_split_image_is_bit_exact_vs_numpy_restatement(ops):
    """aldm_pack_weight + aldm_pack_split_bf16 against oracle/bf16x6.py, bit for bit: layout [k-octet][part][Npad][8]
    and the exact hi/mid/lo truncation split (ragged K and N: zero padding to whole k-tiles / 32 columns)."""
    import numpy as np
    from oracle import bf16x6 as bx
    N, Cin, KH, KW = 40, 12, 2, 3  # K = 72
    w = torch.randn(N, Cin, KH, KW, generator=g(11)) * 0.05
    prev = ops.set_mma("bf16x6")
    try:
        pw = ops.pack_conv(w)
        assert pw.split_ptr() is not None
        img = pw.split.cpu().numpy().view(np.uint16)
    finally:
        ops.set_mma(prev)
    K = Cin * KH * KW
    w_kn = w.permute(2, 3, 1, 0).reshape(K, N).numpy()  # k = (kh*KW + kw)*Cin + ci
    packed = bx.pack_kn(w_kn)
    assert np.array_equal(pw.data.cpu().numpy().reshape(packed.shape), packed)
    ref = bx.split_image(packed, K)
    assert img.size == ref.size and np.array_equal(img.reshape(ref.shape), ref)


def test_two_part_weight_split_image_is_bit_exact_vs_numpy_restatement():
    """aldm_pack_split_bf16_parts(parts = 2) — the "bf16x3" weight image of the DMA-fed kernel — against oracle/bf16x6.py,
    bit for bit: (hi, mid) rounded to nearest even, layout [k-octet][2][Npad][8]."""
    import numpy as np
    from audioldm2_amd import ops as o
    from oracle import bf16x6 as bx
    N, Cin, KH, KW = 40, 12, 2, 3  # K = 72
    w = torch.randn(N, Cin, KH, KW, generator=g(11)) * 0.05
    prev = o.set_mma("bf16x3")
    try:
        pw = o.pack_conv(w)
        assert pw.split_ptr(2) is not None
        img = pw.split2.cpu().numpy().view(np.uint16)
    finally:
        o.set_mma(prev)
    K = Cin * KH * KW
    packed = bx.pack_kn(w.permute(2, 3, 1, 0).reshape(K, N).numpy())
    ref = bx.split_image(packed, K, parts=2)
    assert img.size == ref.size and np.array_equal(img.reshape(ref.shape), ref)


def test_conv_upsample_nearest(ops):
    B, C, H, W = 2, 64, 8, 4
    x = torch.randn(B, C, H, W, generator=g(1))
    w = torch.randn(C, C, 3, 3, generator=g(2)) / math.sqrt(C * 9)
    b = torch.randn(C, generator=g(3))
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    y = ops.conv(cl(x), ops.pack_conv(w, b), pad=(1, 1), up=(2, 2))
    assert rel_err(uncl(y), ref) < gemm_tol()


def test_conv_asymmetric_pad_downsample(ops):
    """VAE encoder Downsample: F.pad (0,1,0,1) then conv k3 s2 p0 (model.py:88-93)."""
    B, C, H, W = 2, 32, 16, 8
    x = torch.randn(B, C, H, W, generator=g(1))
    w = torch.randn(C, C, 3, 3, generator=g(2)) / math.sqrt(C * 9)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, None, stride=2)
    y = ops.conv(cl(x), ops.pack_conv(w), stride=(2, 2), pad=(0, 0), out_hw=(H // 2, W // 2))
    assert rel_err(uncl(y), ref) < gemm_tol()


@pytest.mark.parametrize("M,K,N", [(77, 1024, 640), (4096, 256, 2048), (16, 512, 128), (300, 32, 32),
                                   (1000, 2560, 640)])
def test_linear(ops, M, K, N):
    x = torch.randn(M, K, generator=g(1))
    w = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    b = torch.randn(N, generator=g(3))
    ref = F.linear(x, w, b)
    y = ops.linear(x.cuda(), ops.pack_conv(w, b))
    assert rel_err(y, ref) < gemm_tol()


@pytest.mark.parametrize("C,L,k,d", [(64, 999, 3, 1), (32, 2000, 7, 5), (128, 500, 11, 3)])
def test_conv1d_dilated_resblock_step(ops, C, L, k, d):
    """HiFi-GAN ResBlock inner step x + c(leaky_relu(x)) and the xs/num_kernels accumulation
    (hifigan/models.py:96-103,155-160)."""
    B = 2
    x = torch.randn(B, C, L, generator=g(1))
    w = torch.randn(C, C, k, generator=g(2)) / math.sqrt(C * k)
    b = torch.randn(C, generator=g(3))
    xs0 = torch.randn(B, C, L, generator=g(4))
    pad = (k * d - d) // 2
    r = F.conv1d(F.leaky_relu(x, 0.1), w, b, dilation=d, padding=pad) + x
    ref = xs0 + r / 3
    xl = x.permute(0, 2, 1).contiguous().cuda().view(B, 1, L, C)
    pw = ops.pack_conv(w, b)
    y = ops.conv(xl, pw, pad=(0, pad), dil=(1, d), pre_act=ops.ACT_LRELU, pre_slope=0.1, res=xl)
    assert rel_err(y.view(B, L, C).cpu().permute(0, 2, 1), r) < gemm_tol()
    # accumulate form: out = out + alpha*(conv + bias) + ... checked separately: alpha scales
    # the activation result, residual is added unscaled (see aldm_hip.h epilogue)
    acc = xs0.permute(0, 2, 1).contiguous().cuda().view(B, 1, L, C).clone()
    ops.conv(xl, pw, pad=(0, pad), dil=(1, d), pre_act=ops.ACT_LRELU, pre_slope=0.1, alpha=1.0 / 3,
             out=acc, accumulate=True)
    ref2 = xs0 + (r - x) / 3
    assert rel_err(acc.view(B, L, C).cpu().permute(0, 2, 1), ref2) < gemm_tol()


@pytest.mark.parametrize("Cin,Cout,k,s,L", [(64, 32, 16, 5, 100), (32, 64, 16, 4, 77), (32, 32, 8, 2, 50),
                                            (64, 32, 4, 2, 333), (32, 32, 12, 6, 40), (32, 32, 10, 5, 41)])
def test_conv_transpose1d_polyphase(ops, Cin, Cout, k, s, L):
    """ConvTranspose1d(k, s, padding=(k-s)//2) as s stride-1 convs (hifigan/models.py:127-134)."""
    B = 2
    p = (k - s) // 2
    x = torch.randn(B, Cin, L, generator=g(1))
    w = torch.randn(Cin, Cout, k, generator=g(2)) / math.sqrt(Cin * k / s)
    b = torch.randn(Cout, generator=g(3))
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=p)
    Lout = ref.shape[-1]
    phases = ops.pack_convtr1d(w, b, s)
    T = phases[0].KW
    xl = x.permute(0, 2, 1).contiguous().cuda().view(B, 1, L, Cin)
    out = torch.full((B, 1, Lout, Cout), float("nan"), device="cuda")
    Q = (Lout + p) // s + 2
    for ph in range(s):
        ops.conv(xl, phases[ph], pad=(0, T - 1), out_hw=(1, Q), pre_act=ops.ACT_LRELU, pre_slope=0.1,
                 out=out, remap=(s, ph - p, Lout))
    got = out.view(B, Lout, Cout).cpu().permute(0, 2, 1)
    assert not torch.isnan(got).any(), "polyphase left holes"
    assert rel_err(got, ref) < gemm_tol()


def test_batched_gemm_nt_and_packed(ops):
    """Q K^T and P V of the VAE mid attention (model.py:216-227) through the same engine."""
    Z, M, K, N = 2, 300, 64, 200
    a = torch.randn(Z, M, K, generator=g(1))
    bm = torch.randn(Z, N, K, generator=g(2))
    ref = torch.bmm(a.double(), bm.double().transpose(1, 2)) * 0.125
    y = ops.gemm_nt(a.cuda(), bm.cuda(), alpha=0.125)
    assert rel_err(y, ref) < gemm_tol()
    # P V: [Z, M, Kk] @ [Z, Kk, D]
    Kk, D = 200, 64
    pmat = torch.rand(Z, M, Kk, generator=g(3))
    vmat = torch.randn(Z, Kk, D, generator=g(4))
    ref2 = torch.bmm(pmat.double(), vmat.double())
    y2 = ops.gemm_packed_batched(pmat.cuda(), ops.pack_kn(vmat.cuda()), Kk, D)
    assert rel_err(y2, ref2) < gemm_tol()


def test_frames_gemm_stft(ops):
    """F.conv1d(reflect_pad(x), basis[2F,1,n_fft], stride=hop) (stft.py:58-72)."""
    B, T, n_fft, hop = 2, 4000, 256, 40
    x = torch.rand(B, T, generator=g(1)) - 0.5
    basis = torch.randn(2 * (n_fft // 2 + 1), 1, n_fft, generator=g(2)) / 16
    xp = F.pad(x[:, None, None, :], (n_fft // 2, n_fft // 2, 0, 0), mode="reflect")[:, 0]
    ref = F.conv1d(xp, basis, stride=hop)  # [B, 2F, frames]
    sig = ops.reflect_pad_1d(x.cuda(), n_fft // 2)
    assert rel_err(sig[:, : T + n_fft], xp[:, 0]) == 0.0
    frames = ref.shape[-1]
    pw = ops.pack_conv(basis[:, 0, :])  # [N=2F, K=n_fft] linear layout
    y = ops.frames_gemm(sig, frames, hop, pw)
    assert rel_err(y.cpu().permute(0, 2, 1), ref) < gemm_tol()
    Fq = n_fft // 2 + 1
    mag, ph = ops.mag_phase(y, Fq, 132)
    re, im = ref[:, :Fq], ref[:, Fq:]
    mref = torch.sqrt(re ** 2 + im ** 2).permute(0, 2, 1).reshape(-1, Fq)
    assert rel_err(mag[:, :Fq], mref) < gemm_tol()
    assert float(mag[:, Fq:].abs().max()) == 0.0


@pytest.mark.parametrize("B,P,C,eps", [(2, 4096, 128, 1e-5), (3, 100, 384, 1e-6), (2, 64, 1280, 1e-5),
                                       (1, 1000, 640, 1e-6), (2, 333, 1024, 1e-5)])
def test_groupnorm_stats(ops, B, P, C, eps):
    x = torch.randn(B, P, C, generator=g(1)) * 3 + 1.5
    gamma = torch.randn(C, generator=g(2))
    beta = torch.randn(C, generator=g(3))
    ref = F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, eps=eps).permute(0, 2, 1)
    sc, sh = ops.gn_stats(x.cuda(), gamma.cuda(), beta.cuda(), groups=32, eps=eps)
    got = x * sc.cpu()[:, None, :] + sh.cpu()[:, None, :]
    assert rel_err(got, ref) < 1e-5


@pytest.mark.parametrize("ratio", [1e2, 1e3])
@pytest.mark.parametrize("B,P,C", [(2, 4096, 128), (2, 64, 640), (1, 1000, 256)])
def test_groupnorm_stats_large_mean(ops, ratio, B, P, C):
    """|mean| >> std (a trained checkpoint's post-conv activations): single-pass E[x^2] - E[x]^2 in fp32 loses the
    variance here; the pivot-shifted / Chan-merged statistics must hold 5e-5 against F.group_norm in fp64 (ATen's fp32
    GroupNorm is Welford).  Both the two-launch and the one-block-per-sample (small P*C) forms are covered."""
    x = (torch.randn(B, P, C, generator=g(1)) + ratio * (1 + torch.arange(C) % 7).float() / 4)
    gamma = torch.randn(C, generator=g(2))
    beta = torch.randn(C, generator=g(3))
    ref = F.group_norm(x.double().permute(0, 2, 1), 32, gamma.double(), beta.double(), eps=1e-5).permute(0, 2, 1)
    sc, sh = ops.gn_stats(x.cuda(), gamma.cuda(), beta.cuda(), groups=32, eps=1e-5)
    got = torch.addcmul(sh.cpu().double()[:, None, :], x.double(), sc.cpu().double()[:, None, :])
    assert rel_err(got, ref) < 5e-5
    # the statistics themselves: rstd to 1e-5 relative
    xg = x.double().view(B, P, 32, C // 32).permute(0, 2, 1, 3).reshape(B, 32, -1)
    rstd = (xg.var(-1, unbiased=False) + 1e-5).rsqrt().repeat_interleave(C // 32, 1) * gamma.double()
    assert rel_err(sc, rstd) < 1e-5


def test_groupnorm_stats_constant_channels(ops):
    """A constant input (variance exactly 0): mean exact, rstd = 1/sqrt(eps), no NaN from a negative variance.  The
    engine applies GroupNorm as x*scale + shift (scale = rstd*gamma, shift = beta - mean*scale), so the output error is
    bounded by the fp32 rounding of the two O(|x|*scale) terms: <= 4 * |x| * |scale| * 2^-24."""
    B, P, C = 2, 300, 128
    x = torch.full((B, P, C), 37.25)
    x[1] = -3.0
    gamma = torch.randn(C, generator=g(2))
    beta = torch.randn(C, generator=g(3))
    sc, sh = ops.gn_stats(x.cuda(), gamma.cuda(), beta.cuda(), groups=32, eps=1e-5)
    assert torch.isfinite(sc).all() and torch.isfinite(sh).all()
    assert rel_err(sc, (gamma * 1e-5 ** -0.5).expand(B, C)) < 1e-6
    got = x * sc.cpu()[:, None, :] + sh.cpu()[:, None, :]
    ref = F.group_norm(x.double().permute(0, 2, 1), 32, gamma.double(), beta.double(), eps=1e-5).permute(0, 2, 1)
    bound = 4 * x.abs() * sc.cpu().abs()[:, None, :] * 2.0 ** -24 + 1e-6
    assert ((got.double() - ref).abs() <= bound).all()


@pytest.mark.parametrize("M,C", [(1024, 256), (300, 384), (64, 640), (10, 1280)])
def test_layernorm(ops, M, C):
    x = torch.randn(M, C, generator=g(1)) * 2 + 0.3
    gamma = torch.randn(C, generator=g(2))
    beta = torch.randn(C, generator=g(3))
    ref = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    y = ops.layernorm(x.cuda(), gamma.cuda(), beta.cuda(), 1e-5)
    assert rel_err(y, ref) < 5e-6


ATTN_MODE_NAME = {1: "f32", 2: "bf16x6", 3: "bf16x3"}   # aldm_attention_mma codes


def ref_attention(q, k, v, heads, mask=None):
    """attention.py:343-367 restated (einsum / masked_fill(-finfo.max) / softmax / einsum)."""
    B, Lq, Cc = q.shape
    d = Cc // heads
    def sp(t):
        return t.view(B, -1, heads, d).permute(0, 2, 1, 3).reshape(B * heads, -1, d)
    qh, kh, vh = sp(q), sp(k), sp(v)
    sim = torch.einsum("bid,bjd->bij", qh, kh) * d ** -0.5
    if mask is not None:
        m = mask.reshape(B, -1)[:, None, :].repeat_interleave(heads, 0)
        sim = sim.masked_fill(~(m == 1), -torch.finfo(sim.dtype).max)
    att = sim.softmax(-1)
    o = torch.einsum("bij,bjd->bid", att, vh)
    return o.view(B, heads, Lq, d).permute(0, 2, 1, 3).reshape(B, Lq, Cc)


@pytest.mark.parametrize("B,heads,Lq,Lk,masked", [
    (2, 8, 1024, 1024, False),   # UNet level-1 self attention
    (2, 12, 256, 256, False),
    (3, 20, 64, 64, False),
    (2, 8, 1024, 8, True),       # cross attention to 8 AudioMAE tokens
    (2, 12, 256, 77, True),      # ragged Lk, partial mask
    (2, 4, 100, 33, True),       # ragged Lq and Lk
    (2, 4, 100, 64, True),       # ragged Lq, full key tiles: the software-pipelined kernel with clamped query rows
    (2, 4, 70, 96, False),       # three key tiles, a second (partial) wave of queries
    (16, 8, 64, 32, True),       # 64 queries per wave on a 64-query sample (three of the block's four waves exit)
    (1, 2, 40, 1, False),        # single key (uncond T5 token)
])
def test_attention_d32(ops, B, heads, Lq, Lk, masked):
    Cc = heads * 32
    # fused-QKV style buffers: q/k/v are column slices of wider row-major buffers
    qb = torch.randn(B, Lq, Cc + 64, generator=g(1))
    kvb = torch.randn(B, Lk, 2 * Cc, generator=g(2))
    q, k, v = qb[:, :, :Cc], kvb[:, :, :Cc], kvb[:, :, Cc:]
    mask = None
    if masked:
        mask = (torch.rand(B, Lk, generator=g(3)) > 0.3).float()
        mask[0, :] = 0  # sample 0: every key masked -> reference degenerates to uniform weights
        if B > 1:
            mask[1, 0] = 1
    ref = ref_attention(q.double().contiguous(), k.double().contiguous(), v.double().contiguous(), heads, mask)   # fp64
    qd, kvd = qb.cuda(), kvb.cuda()
    for mode in (1, 2, 3):  # fp32 MFMA, bf16x6, bf16x3
        prev = ops.attention_mma(mode)
        try:
            y = ops.attention(qd[:, :, :Cc], kvd[:, :, :Cc], kvd[:, :, Cc:], heads,
                              mask=None if mask is None else mask.cuda())
        finally:
            ops.attention_mma(prev)
        assert rel_err(y, ref) < fused_tol(ATTN_MODE_NAME[mode]), f"attention mma mode {mode}"


@pytest.mark.parametrize("Lk,masked", [(8, True), (8, False), (40, True), (33, False), (32, True), (64, False), (1, False),
                                       (96, True)])
def test_attention_loads_no_key_past_the_last(ops, Lk, masked):
    """K / V are the first B*Lk rows of a buffer whose remainder is NaN: a V row >= Lk of the last sample entering the
    product shows up as NaN (0 * NaN) — and at the end of a mapped segment it is a page fault, which is how the bug this
    guards against surfaced (an abort in the model tests, not in any unit test).  Every kernel variant prefetches the tile
    after the last one and the ragged ones address keys per lane: all of that has to stay below Lk."""
    B, heads, Lq = 2, 8, 256
    Cc = heads * 32
    qd = torch.randn(B, Lq, Cc, generator=g(1)).cuda()
    flat = torch.full((B * Lk * 2 * Cc + 64 * 2 * Cc,), float("nan"))
    flat[: B * Lk * 2 * Cc] = torch.randn(B * Lk * 2 * Cc, generator=g(2))
    flat = flat.cuda()
    kvd = flat[: B * Lk * 2 * Cc].view(B, Lk, 2 * Cc)
    mask = None
    if masked:
        mask = torch.ones(B, Lk)
        mask[:, Lk // 2:] = 0
        mask[:, 0] = 1
        mask = mask.cuda()
    ref = ref_attention(qd.cpu().double(), kvd[:, :, :Cc].cpu().double().contiguous(), kvd[:, :, Cc:].cpu().double().contiguous(),
                        heads, None if mask is None else mask.cpu())
    for mode in (1, 2, 3):
        prev = ops.attention_mma(mode)
        try:
            y = ops.attention(qd, kvd[:, :, :Cc], kvd[:, :, Cc:], heads, mask=mask)
        finally:
            ops.attention_mma(prev)
        assert torch.isfinite(y).all(), f"mode {mode}: a key past Lk entered the product"
        assert rel_err(y, ref) < fused_tol(ATTN_MODE_NAME[mode]), f"mode {mode}"


def test_attention_online_softmax_rescale(ops):
    """Force the running-max rescale branch: one late key dominates (guide rule 26)."""
    B, heads, L = 1, 1, 256
    q = torch.randn(B, L, 32, generator=g(1))
    k = torch.randn(B, L, 32, generator=g(2))
    v = torch.randn(B, L, 32, generator=g(3))
    k[0, 200] = q[0, 5] * 20.0   # spike: query 5 against key 200 (7th tile)
    k[0, 3] = q[0, 100] * 15.0   # and an early spike for another row
    ref = ref_attention(q.double(), k.double(), v.double(), heads)
    y = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads)
    assert rel_err(y, ref) < fused_tol()   # the attention kernel follows the library's product mode (bf16x6 by default)


def test_softmax_rows_geglu_embedding(ops):
    x = torch.randn(64, 4096, generator=g(1)) * 4
    assert rel_err(ops.softmax_rows(x.cuda(), 0.044), (x * 0.044).softmax(-1)) < 1e-5
    xg = torch.randn(50, 2 * 1024, generator=g(2)) * 2
    a, gate = xg.chunk(2, -1)
    assert rel_err(ops.geglu(xg.cuda()), a * F.gelu(gate)) < 1e-5
    t = torch.tensor([1.0, 6.0, 501.0, 996.0])
    half = 64
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None] * freqs[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], -1)
    got = ops.timestep_embedding(t.cuda(), 128)
    assert float((got.cpu() - ref).abs().max()) < 2e-4  # sin/cos of arguments up to ~1e3 rad


def test_layout_and_ddim_step(ops):
    x = torch.randn(3, 8, 20, 16, generator=g(1))
    y = ops.nchw_to_nhwc(x.cuda(), rep=2)
    assert torch.equal(y.cpu()[:3], x.permute(0, 2, 3, 1)) and torch.equal(y.cpu()[3:], x.permute(0, 2, 3, 1))
    assert torch.equal(ops.nhwc_to_nchw(y[:3].contiguous()).cpu(), x)
    # ddim.py:298-355 restated with torch ops
    eu = torch.randn(3, 8, 20, 16, generator=g(2))
    ec = torch.randn(3, 8, 20, 16, generator=g(3))
    nz = torch.randn(3, 8, 20, 16, generator=g(4))
    a_t, a_prev, sig, s = torch.tensor(0.37), torch.tensor(0.41), torch.tensor(0.12), 3.5
    e = eu + s * (ec - eu)
    pred = (x - (1 - a_t).sqrt() * e) / a_t.sqrt()
    xp = a_prev.sqrt() * pred + (1.0 - a_prev - sig ** 2).sqrt() * e + sig * nz
    coef = torch.tensor([float((1 - a_t).sqrt()), float(a_t.sqrt()),
                         float((1.0 - a_prev - sig ** 2).sqrt()), float(a_prev.sqrt()), float(sig),
                         s, 1.0, 0.0]).cuda()
    gx, gp = ops.ddim_step(x.cuda(), torch.stack([eu, ec]).cuda(), nz.cuda(), coef)
    assert rel_err(gx, xp) < EW_TOL and rel_err(gp, pred) < EW_TOL


def test_ddpm_step_and_inpaint_blend(ops):
    """Ancestral update (ddpm.py:357-373, 1127-1181) and the inpainting blend (ddim.py:226-231)."""
    x = torch.randn(2, 8, 12, 16, generator=g(1))
    e = torch.randn(2, 8, 12, 16, generator=g(2))
    nz = torch.randn(2, 8, 12, 16, generator=g(3))
    a, b, c1, c2, sg = 1.31, 0.85, 0.07, 0.93, 0.11
    x0 = a * x - b * e
    ref = (c1 * x0 + c2 * x) + sg * nz
    got = ops.ddpm_step(x.cuda(), e.cuda(), nz.cuda(), torch.tensor([a, b, c1, c2, sg, 0, 0, 0]).cuda())
    assert rel_err(got, ref) < EW_TOL
    mask = (torch.rand(2, 8, 12, 16, generator=g(4)) > 0.5).float()
    xs = torch.randn(2, 8, 12, 16, generator=g(5))
    sa, so = 0.6, 0.8
    ref2 = (sa * xs + so * nz) * mask + (1.0 - mask) * x
    xc = x.clone().cuda()
    ops.inpaint_blend(xc, xs.cuda(), nz.cuda(), mask.cuda(), torch.tensor([sa, so]).cuda())
    assert rel_err(xc, ref2) < EW_TOL


def test_error_reporting(ops):
    """Errors surface as RuntimeError with the library's message (no silent fallback)."""
    x = torch.randn(1, 4, 4, 6).cuda()  # C = 6 is not a multiple of 4
    w = torch.randn(8, 6, 1, 1)
    with pytest.raises(RuntimeError, match="multiples of 4"):
        ops.conv(x, ops.pack_conv(w))


def test_env_selected_strict_mode_reaches_the_attention_kernel():
    """`ALDM_MMA=bf16x6` (the documented strict mode) in the ENVIRONMENT must put the attention kernel on bf16x6
    as well, not only ops.set_mma(): a fresh process with the variable set gives fp32-grade attention (error vs fp64 ~2e-7;
    the bf16x3 default sits at ~4e-6 on these inputs)."""
    import os
    import subprocess
    import sys
    code = r"""
import torch, torch.nn.functional as F
from audioldm2_amd import ops
g = torch.Generator().manual_seed(7)
B, H, L = 2, 8, 256
q, k, v = (torch.randn(B, L, H * 32, generator=g) for _ in range(3))
o = ops.attention(q.cuda(), k.cuda(), v.cuda(), H).double().cpu()
sh = lambda t: t.double().view(B, L, H, 32).transpose(1, 2)
ref = F.scaled_dot_product_attention(sh(q), sh(k), sh(v)).transpose(1, 2).reshape(B, L, H * 32)
print("ERR", float((o - ref).abs().max() / ref.abs().max()))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    errs = {}
    for mode in ("bf16x6", "bf16x3"):
        env = dict(os.environ, ALDM_MMA=mode, PYTHONPATH=root)
        env.pop("ALDM_ATTN_MMA", None)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        errs[mode] = float([l for l in out.stdout.splitlines() if l.startswith("ERR")][0].split()[1])
    assert errs["bf16x6"] < 1e-6, errs
    assert errs["bf16x3"] > 2 * errs["bf16x6"], errs   # the two modes really are different kernels


# ---- bf16x3 under trained-like statistics (VERDICT r2 "weak" #2, next #3d) ----------------------------------------------------
# Every other test feeds the kernels U(+-1/sqrt(fan_in)) weights and O(1) activations; a trained checkpoint has heavy-tailed weights,
# channels with |mean| >> std in front of the skip convs and attention logits tens of units apart.  The per-mode stress bars
# (max-norm relative error against fp64 of the SAME fp32 inputs: 5e-6 for the fp32-grade default, 5e-5 for the opt-in bf16x3 mode
# with its 16 significant bits per operand) must hold there too.
def report(line):
    """stdout + gpurun_out/parity_report.txt (merged back by gpurun)"""
    import os
    print(line)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_report.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def _heavy(shape, gen):
    """Student-t(3) samples (seeded): heavy tails, |x| up to tens of standard deviations."""
    z = torch.randn(shape, generator=gen)
    chi = torch.stack([torch.randn(shape, generator=gen) ** 2 for _ in range(3)]).sum(0)
    return z / torch.sqrt(chi / 3.0)


@pytest.mark.parametrize("mode", ["bf16x3", "bf16x6"])
def test_split_gemm_heavy_tailed_weights_and_large_mean_activations(mode):
    from audioldm2_amd import ops as o
    prev = o.set_mma(mode)
    try:
        gen = torch.Generator().manual_seed(11)
        M, K, N = 2048, 640, 384
        w = _heavy((N, K), gen) / math.sqrt(K)                 # |w| up to ~30x its rms
        x = torch.randn(1, M, K, generator=gen)
        x[..., ::7] += 1.0e3                                    # every 7th channel: mean / std = 1e3
        x[..., 3::11] *= 50.0                                   # outlier channels
        b = torch.randn(N, generator=gen)
        xs = o.split_rows(x.cuda())                             # the product path: pre-split operand, DMA-fed GEMM
        y = o.linear(xs, o.pack_conv(w, b))
        ref = x.double() @ w.double().t() + b.double()
        err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
        report(f"stress GEMM heavy-tailed W, mean/std 1e3 channels ({mode}): max-norm rel err {err:.2e}")
        assert log_err(err, stress_tol(mode), "stress gemm") < stress_tol(mode), err   # per-mode: 5e-6 (measured 1.1e-6) / 5e-5 (5.6e-6)
    finally:
        o.set_mma(prev)


@pytest.mark.parametrize("mode", ["bf16x3", "bf16x6"])
def test_attention_sharp_logits(mode):
    """Scores spread over +-30 (softmax close to one-hot with a few competing keys): the exponentials amplify any error of the
    score products."""
    from audioldm2_amd import ops as o
    prev = o.set_mma(mode)
    try:
        gen = torch.Generator().manual_seed(5)
        B, H, L = 2, 8, 512
        q = torch.randn(B, L, H * 32, generator=gen) * 4.0
        k = torch.randn(B, L, H * 32, generator=gen) * 4.0     # q.k / sqrt(32) ~ N(0, 16^2 / ...): |logit| reaches 30+
        v = _heavy((B, L, H * 32), gen)
        out = o.attention(q.cuda(), k.cuda(), v.cuda(), H).double().cpu()
        sh = lambda t: t.double().view(B, L, H, 32).transpose(1, 2)
        s = sh(q) @ sh(k).transpose(-1, -2) / math.sqrt(32)
        assert float(s.abs().max()) > 30.0
        ref = (torch.softmax(s, -1) @ sh(v)).transpose(1, 2).reshape(B, L, H * 32)
        err = float((out - ref).abs().max() / ref.abs().max())
        report(f"stress attention |logit| up to {float(s.abs().max()):.0f} ({mode}): max-norm rel err {err:.2e}")
        assert log_err(err, stress_tol(mode), "stress attention") < stress_tol(mode), err   # 5e-6 (measured 7.0e-7) / 5e-5 (1.4e-5)
    finally:
        o.set_mma(prev)
