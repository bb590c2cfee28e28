"""DMA-fed bf16-split GEMM over pre-split operands (csrc/igemm_dma.h) and the split-image producers, through the C ABI,
against PyTorch on the CPU evaluated in fp64.  Per-mode tolerances (tests/tolerances.py): bf16x6 contractions max|err|/max|ref| <=
2e-6 (5e-6 around a GELU / SiLU / softmax), bf16x3 <= 5e-5 — and test_five_product_gemm_fails_the_fp32_grade_bar shows that the
bf16x6 bar catches a kernel that loses ONE of the six partial products; the split itself is exact (hi + mid + lo == x bitwise)."""
import math

import pytest
import torch
from tolerances import F64 as F   # references in fp64 (every floating argument promoted)
from tolerances import fused_tol, gemm_tol, log_err, stress_tol

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


@pytest.fixture(scope="module", params=["bf16x6", "bf16x3"])
def ops(request):
    """Both split modes of the DMA-fed kernel: "bf16x6" (exact 3-part images, 6 partial products: held to the fp32-grade bar) and
    "bf16x3" ((hi, mid) rounded to nearest, 3 partial products: its 2^-17 operand rounding shows as 4e-6 .. 1.7e-5, held to 5e-5)."""
    from audioldm2_amd import ops as o
    prev = o.set_mma(request.param)
    yield o
    o.set_mma(prev)


def exact_split(ops):
    return ops.split_parts() == 3


def assert_split_equals(ops, s, ref, what=""):
    """A 3-part image reproduces the fp32 values bitwise; a 2-part image to 2^-17 relative (round to nearest twice)."""
    if exact_split(ops):
        assert torch.equal(s.float(), ref), what
    else:
        err = (s.float().double() - ref.double()).abs()
        assert bool((err <= ref.double().abs() * 2.0 ** -17 + 1e-38).all()), what


def test_split_rows_is_exact_and_applies_groupnorm_silu(ops):
    B, P, C1, C2 = 3, 50, 96, 64
    x1 = (torch.randn(B, P, C1, generator=g(1)) * 3 + 1).cuda()
    x2 = torch.randn(B, P, C2, generator=g(2)).cuda() * 1e-3
    sc = torch.randn(B, C1 + C2, generator=g(3)).cuda()
    sh = torch.randn(B, C1 + C2, generator=g(4)).cuda()
    s, raw = ops.split_rows(x1, x2, pre=(sc, sh), act=ops.ACT_SILU, want_raw=True)
    xc = torch.cat([x1, x2], -1)
    assert_split_equals(ops, raw, xc, "hi + mid (+ lo) must reproduce x")
    ref = F.silu(xc * sc[:, None, :] + sh[:, None, :])
    assert rel_err(s.float(), ref) < (2e-6 if exact_split(ops) else 2e-5)   # 2-part image: 2^-16 per value
    s2 = ops.split_rows(x1, pre=(sc[:, :C1].contiguous(), sh[:, :C1].contiguous()))
    assert rel_err(s2.float(), x1.double() * sc[:, None, :C1].double() + sh[:, None, :C1].double()) < \
        (2e-7 if exact_split(ops) else 2e-5)  # one fma rounding (+ the 2^-16 of a 2-part image)
    # tiny / huge values keep the exact split too
    z = torch.tensor([1e-25, -3e38, 1.0000001, -0.0, 65504.0, 1e-30, 7.0, 3.14159] * 4).view(1, 1, 32).cuda()
    assert_split_equals(ops, ops.split_rows(z), z)


@pytest.mark.parametrize("B,C,N,H,W,k,s,p,up", [
    (2, 128, 128, 32, 16, 3, 1, 1, 1),     # UNet level-0 ResBlock conv
    (1, 128, 128, 300, 16, 3, 1, 1, 1),    # M = 4800: ragged M tile
    (2, 256, 256, 16, 8, 3, 2, 1, 1),      # Downsample stride 2
    (2, 256, 256, 8, 4, 3, 1, 1, 2),       # Upsample: nearest x2 in the address generation
    (2, 640, 640, 4, 2, 3, 1, 1, 1),       # deepest level, tiny M
    (2, 256, 384, 9, 5, 1, 1, 0, 1),       # 1x1 conv, odd extents
    (1, 64, 96, 50, 30, 3, 1, 1, 1),       # N = 96: partial column tile
    (3, 32, 40, 7, 5, 3, 1, 2, 1),         # one k-tile per tap, padding 2, N % 32 != 0
])
def test_conv_on_split_operand(ops, B, C, N, H, W, k, s, p, up):
    x = torch.randn(B, C, H, W, generator=g(1))
    w = torch.randn(N, C, k, k, generator=g(2)) / math.sqrt(C * k * k)
    b = torch.randn(N, generator=g(3))
    xin = F.interpolate(x, scale_factor=up, mode="nearest") if up > 1 else x
    ref = F.conv2d(xin, w, b, stride=s, padding=p)
    pw = ops.pack_conv(w, b)
    xs = ops.split_rows(cl(x))
    y = ops.conv(xs, pw, stride=(s, s), pad=(p, p), up=(up, up))
    assert rel_err(uncl(y), ref) < gemm_tol()


_TILES = {"bf16x6": [(256, 128, 2), (128, 128, 3), (128, 128, 2), (64, 128, 4), (64, 128, 2), (128, 64, 4), (128, 64, 2),
                     (64, 64, 3), (64, 64, 2)],
          "bf16x3": [(256, 128, 2), (256, 128, 3), (128, 128, 4), (128, 128, 2), (64, 128, 6), (64, 128, 4), (64, 128, 2),
                     (128, 64, 6), (128, 64, 4), (128, 64, 2), (64, 64, 6), (64, 64, 3), (64, 64, 2)]}


@pytest.mark.parametrize("mode,bm,bn,st", [(m, *t) for m, ts in _TILES.items() for t in ts])
@pytest.mark.parametrize("splits", [1, 3])
def test_dma_every_tile_and_splitk(mode, bm, bn, st, splits):
    """Every instantiation of both split modes, with and without split-K, ragged M, K = 36 k-tiles (ragged split), full
    epilogue, and the split-image second output equal to the fp32 one."""
    from audioldm2_amd import ops
    prev = ops.set_mma(mode)
    try:
        _every_tile(ops, bm, bn, st, splits)
    finally:
        ops.set_mma(prev)


def _every_tile(ops, bm, bn, st, splits):
    B, C, N, H, W = 3, 128, 96, 13, 7
    x = torch.randn(B, C, H, W, generator=g(1))
    w = torch.randn(N, C, 3, 3, generator=g(2)) / math.sqrt(C * 9)
    b = torch.randn(N, generator=g(3))
    emb = torch.randn(B, 2 * N, generator=g(4))
    res = torch.randn(B, N, H, W, generator=g(5))
    ref = F.silu(F.conv2d(x, w, b, padding=1) + emb[:, N:, None, None]) + res
    pw = ops.pack_conv(w, b)
    xs = ops.split_rows(cl(x))
    ops.igemm_force(bm, bn, splits, 0, st)
    try:
        y1, s1 = ops.conv(xs, pw, pad=(1, 1), rowbias=emb.cuda()[:, N:], act=ops.ACT_SILU, res=cl(res), split_out="also")
        y2 = ops.conv(xs, pw, pad=(1, 1), rowbias=emb.cuda()[:, N:], act=ops.ACT_SILU, res=cl(res))
        s3 = ops.conv(xs, pw, pad=(1, 1), rowbias=emb.cuda()[:, N:], act=ops.ACT_SILU, res=cl(res), split_out="only")
    finally:
        ops.igemm_force(0, 0, 0)
    assert rel_err(uncl(y1), ref) < fused_tol()
    assert torch.equal(y1, y2), "must be bitwise reproducible"
    assert_split_equals(ops, s1, y1)
    assert_split_equals(ops, s3, y1)


_FIVE_PRODUCT_SCRIPT = r"""
import math, sys, torch
from audioldm2_amd import lib, ops
sys.path.insert(0, {tests!r})
from tolerances import fused_tol, gemm_tol
rel_err = lambda a, b: float((a.detach().double().cpu() - b).abs().max() / b.abs().max())
assert lib.LIB_PATH.endswith("libaldm_hip_testhooks.so"), lib.LIB_PATH
g = lambda s: torch.Generator().manual_seed(s)
prev = ops.set_mma("bf16x6")
M, K, N = 4096, 640, 384
x = torch.randn(1, M, K, generator=g(1))
w = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
b = torch.randn(N, generator=g(3))
ref = x.double() @ w.double().t() + b.double()
xs, pw = ops.split_rows(x.cuda()), ops.pack_conv(w, b)
ops.igemm_force(64, 128, 1, 0, 2)
e6 = rel_err(ops.linear(xs, pw), ref)
assert not ops.debug_drop_product(True)
e5 = rel_err(ops.linear(xs, pw), ref)
ops.igemm_force(128, 128, 1, 0, 3)          # no five-product form of this tile: must fail, not run
try:
    ops.linear(xs, pw)
    raise SystemExit("a launch without a five-product form ran while the switch was on")
except RuntimeError as e:
    assert "aldm_debug_drop_product" in str(e), str(e)
assert ops.debug_drop_product(False)
ops.igemm_force(64, 128, 1, 0, 2)
e6b = rel_err(ops.linear(xs, pw), ref)      # switch off again: full precision
ops.igemm_force(0, 0, 0)
print(f"RESULT {{e6:.6e}} {{e5:.6e}} {{e6b:.6e}} {{gemm_tol('bf16x6'):.3e}} {{fused_tol('bf16x6'):.3e}}")
"""


def test_five_product_gemm_fails_the_fp32_grade_bar():
    """the bf16x6 bars must be able to tell the credited mode from a narrower one.  A VARIANT of the library
    (libaldm_hip_testhooks.so: the same sources under -DALDM_TEST_HOOKS, built next to the release library; VERDICT r5 next #8 —
    the release libaldm_hip.so no longer carries the switch, ABI v9) has ONE deliberately broken instantiation
    (aldm_debug_drop_product: the classic 64x128 tile without its smallest partial product, hi_a x lo_w).  A subprocess loads the
    variant through $ALDM_LIB_PATH: on the same launch the six-product kernel meets gemm_tol("bf16x6") = 2e-6 against fp64 and the
    five-product kernel misses it (and the 5e-6 fused / stress bar) by a wide margin, while it would have sailed through the old 5e-5
    bar; with the switch on, any launch that has no five-product form fails instead of silently running at full precision."""
    import os
    import subprocess
    import sys
    from audioldm2_amd import lib, ops
    assert not hasattr(lib.load(), "aldm_debug_drop_product"), "the release library must not export the test hook"
    with pytest.raises(RuntimeError, match="test hook"):
        ops.debug_drop_product(True)
    assert os.path.exists(lib.TESTHOOKS_LIB_PATH), "build() also builds audioldm2_amd/libaldm_hip_testhooks.so"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ALDM_LIB_PATH=lib.TESTHOOKS_LIB_PATH, PYTHONPATH=root)
    env.pop("ALDM_ERR_LOG", None)
    out = subprocess.run([sys.executable, "-c", _FIVE_PRODUCT_SCRIPT.format(tests=os.path.join(root, "tests"))], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    e6, e5, e6b, gt, ft = (float(v) for v in [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()[1:])
    print(f"six products {e6:.2e}, five products {e5:.2e}, six again {e6b:.2e} (bars: gemm {gt:.0e}, fused {ft:.0e})")
    assert e6 < gemm_tol("bf16x6") and e6b < gemm_tol("bf16x6")
    # exact arithmetic puts the lost product at 1.0e-5 of max|ref| here (truncation splits: lo_w has the sign of w, the loss is coherent)
    assert e5 > fused_tol("bf16x6") and e5 < 5e-5, "the five-product kernel passes the OLD 5e-5 bar and fails every fp32-grade one"


def test_dma_matches_register_staged_kernel_bitwise_class(ops):
    """Same arithmetic as the register-staged bf16-split kernel: results agree to fp32 summation-order noise."""
    B, C, N, H, W = 2, 256, 256, 16, 8
    x = torch.randn(B, C, H, W, generator=g(1))
    w = torch.randn(N, C, 3, 3, generator=g(2)) / math.sqrt(C * 9)
    pw = ops.pack_conv(w, None)
    y_old = ops.conv(cl(x), pw, pad=(1, 1))
    y_new = ops.conv(ops.split_rows(cl(x)), pw, pad=(1, 1))
    assert rel_err(y_new, y_old) < (2e-6 if exact_split(ops) else 1e-5)


def test_linear_geglu_layernorm_attention_split_chain(ops):
    """LayerNorm -> (split) -> GEGLU projection -> (split) -> FF out projection, and attention -> (split) -> out
    projection: the transformer block's GEMM chain with no fp32 operand between producer and consumer."""
    M, C = 300, 256
    x = torch.randn(2, M // 2, C, generator=g(1))
    ga, be = torch.randn(C, generator=g(2)), torch.randn(C, generator=g(3))
    w1 = torch.randn(8 * C, C, generator=g(4)) / math.sqrt(C)
    b1 = torch.randn(8 * C, generator=g(5))
    w2 = torch.randn(C, 4 * C, generator=g(6)) / math.sqrt(4 * C)
    b2 = torch.randn(C, generator=g(7))
    n_ref = F.layer_norm(x, (C,), ga, be, 1e-5)
    h = n_ref @ w1.double().t() + b1.double()
    a, gate = h.chunk(2, -1)
    ff_ref = (a * F.gelu(gate)) @ w2.double().t() + b2.double() + x.double()
    n_f, n_s = ops.layernorm(x.cuda(), ga.cuda(), be.cuda(), split_out="also")
    assert rel_err(n_f, n_ref) < 2e-6
    assert_split_equals(ops, n_s, n_f)
    gs = ops.linear_geglu(n_s, ops.pack_geglu(w1, b1), split_out="only")
    y = ops.linear(gs, ops.pack_conv(w2, b2), res=x.cuda())
    assert rel_err(y, ff_ref) < fused_tol()
    # attention with a split-image output
    heads, Lq, Lk = 8, 70, 45
    q = torch.randn(2, Lq, heads * 32, generator=g(8))
    k = torch.randn(2, Lk, heads * 32, generator=g(9))
    v = torch.randn(2, Lk, heads * 32, generator=g(10))
    o_f, o_s = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, split_out="also")
    assert_split_equals(ops, o_s, o_f)
    o_only = ops.attention(q.cuda(), k.cuda(), v.cuda(), heads, split_out="only")
    assert_split_equals(ops, o_only, o_f)
    qh = q.view(2, Lq, heads, 32).transpose(1, 2)
    kh = k.view(2, Lk, heads, 32).transpose(1, 2)
    vh = v.view(2, Lk, heads, 32).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(2, Lq, heads * 32)
    assert rel_err(o_f, ref) < fused_tol()


# ---- the persistent wave-specialised form (csrc/igemm_dma_ws.h): aldm_igemm_force_stages(100 + ring depth) -------------
_WS_TILES = {"bf16x6": [(64, 128, 2), (64, 128, 3), (128, 64, 2), (128, 64, 3), (64, 64, 2), (64, 64, 3), (64, 64, 4)],
             "bf16x3": [(64, 128, 2), (64, 128, 3), (64, 128, 4), (64, 128, 5), (128, 64, 2), (128, 64, 3), (128, 64, 4),
                        (64, 64, 3), (64, 64, 4), (64, 64, 6)]}
_OLD_STAGES = {(64, 128): 4, (128, 64): 4, (64, 64): 3}


def _ws_vs_old(ops, bm, bn, st, fn):
    """fn() under the persistent kernel and under igemm_dma_kernel on the same tile: same K order, same epilogue order ->
    the results must agree BITWISE."""
    ops.igemm_force(bm, bn, 1, 0, 100 + st)
    try:
        y_ws = fn()
    finally:
        ops.igemm_force(0, 0, 0)
    ops.igemm_force(bm, bn, 1, 0, _OLD_STAGES[(bm, bn)])
    try:
        y_old = fn()
    finally:
        ops.igemm_force(0, 0, 0)
    return y_ws, y_old


@pytest.mark.parametrize("mode,bm,bn,st", [(m, *t) for m, ts in _WS_TILES.items() for t in ts])
def test_dma_ws_every_tile(mode, bm, bn, st):
    """Every instantiation of the persistent kernel: a conv with a timestep row bias, a conv with a residual and both outputs,
    with MORE tiles than compute units (each block walks several tiles: hand-over, prefetch one tile ahead, ring restart)."""
    from audioldm2_amd import ops
    prev = ops.set_mma(mode)
    try:
        B, C, N, H, W = 5, 64, 128, 64, 64          # M = 20480 rows = 320 row tiles of 64; every sample a whole number of tiles
        x = torch.randn(B, C, H, W, generator=g(1))
        w = torch.randn(N, C, 3, 3, generator=g(2)) / math.sqrt(C * 9)
        b = torch.randn(N, generator=g(3))
        emb = torch.randn(B, 2 * N, generator=g(4))
        res = torch.randn(B, N, H, W, generator=g(5))
        pw = ops.pack_conv(w, b)
        xs = ops.split_rows(cl(x))
        conv = F.conv2d(x, w, b, padding=1)
        y_ws, y_old = _ws_vs_old(ops, bm, bn, st, lambda: ops.conv(xs, pw, pad=(1, 1), rowbias=emb.cuda()[:, N:]))
        assert rel_err(uncl(y_ws), conv + emb[:, N:, None, None]) < gemm_tol()
        assert torch.equal(y_ws, y_old)
        (y_ws, s_ws), (y_old, s_old) = _ws_vs_old(
            ops, bm, bn, st, lambda: ops.conv(xs, pw, pad=(1, 1), res=cl(res), alpha=0.5, split_out="also"))
        assert rel_err(uncl(y_ws), 0.5 * (conv + res)) < gemm_tol()
        assert torch.equal(y_ws, y_old) and torch.equal(s_ws.data, s_old.data)
        assert_split_equals(ops, s_ws, y_ws)
        s_only, _ = _ws_vs_old(ops, bm, bn, st, lambda: ops.conv(xs, pw, pad=(1, 1), res=cl(res), alpha=0.5, split_out="only"))
        assert torch.equal(s_only.data, s_ws.data)
    finally:
        ops.set_mma(prev)


@pytest.mark.parametrize("K", [32, 64, 96, 256, 1024])
@pytest.mark.parametrize("M", [64 * 4, 64 * 700])
def test_dma_ws_linear_k_edges(ops, K, M):
    """K = 32 / 64: zero / one in-loop barrier per tile (the two roles must still agree on the barrier sequence); K shorter
    than the ring; M = 256 rows: fewer tiles than compute units (one tile per block, some blocks absent)."""
    N = 256
    x = torch.randn(1, M, K, generator=g(1))
    w = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    b = torch.randn(N, generator=g(3))
    res = torch.randn(1, M, N, generator=g(4))
    pw = ops.pack_conv(w, b)
    xs = ops.split_rows(x.cuda())
    ref = xs.float().double().cpu() @ w.double().t() + b.double() + res.double()
    for bm, bn, st in _WS_TILES["bf16x3" if ops.split_parts() == 2 else "bf16x6"][::2]:
        y_ws, y_old = _ws_vs_old(ops, bm, bn, st, lambda: ops.linear(xs, pw, res=res.cuda()))
        assert rel_err(y_ws, ref) < gemm_tol(), (bm, bn, st)
        assert torch.equal(y_ws, y_old), (bm, bn, st)


def test_dma_ws_geglu(ops):
    """The GEGLU epilogue of the persistent kernel (128-column tiles), split-image and fp32 outputs."""
    M, C = 64 * 300, 128
    x = torch.randn(1, M, C, generator=g(1))
    w1 = torch.randn(8 * C, C, generator=g(4)) / math.sqrt(C)
    b1 = torch.randn(8 * C, generator=g(5))
    pw = ops.pack_geglu(w1, b1)
    xs = ops.split_rows(x.cuda())
    h = xs.float().double().cpu() @ w1.double().t() + b1.double()
    a, gate = h.chunk(2, -1)
    ref = a * F.gelu(gate)
    for bm, bn, st in [t for t in _WS_TILES["bf16x3" if ops.split_parts() == 2 else "bf16x6"] if t[1] == 128]:
        (y_ws, s_ws), (y_old, s_old) = _ws_vs_old(ops, bm, bn, st, lambda: ops.linear_geglu(xs, pw, split_out="also"))
        assert rel_err(y_ws, ref) < fused_tol()
        assert torch.equal(y_ws, y_old) and torch.equal(s_ws.data, s_old.data)


def test_dma_ws_falls_back_when_not_eligible(ops):
    """A forced persistent launch the kernel cannot run (ragged M) fails loudly; a HINTED one (tuned tables are keyed by
    geometry only) quietly stays on igemm_dma_kernel."""
    M, K, N = 100, 64, 128
    x = torch.randn(1, M, K, generator=g(1))
    w = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    pw = ops.pack_conv(w, None)
    xs = ops.split_rows(x.cuda())
    st = 100 + (4 if ops.split_parts() == 2 else 3)
    ops.igemm_force(64, 128, 1, 0, st)
    try:
        with pytest.raises(RuntimeError):
            ops.linear(xs, pw)
    finally:
        ops.igemm_force(0, 0, 0)


@pytest.mark.parametrize("B,P,C1,C2,act", [
    (16, 1024, 256, 0, True),     # UNet level 1, 16 samples: two groups per block, one launch
    (16, 256, 384, 384, True),    # skip concat at level 2
    (16, 64, 640, 0, False),      # SpatialTransformer.norm at level 3 (no activation)
    (16, 1024, 384, 128, True),   # concat with a group straddling ... the x1 / x2 seam stays piece aligned (C1 % 8 == 0)
    (2, 1024, 128, 0, True),      # batch 2: one group per block, slab of 4 channels -> the two-launch form
    (3, 4096, 128, 128, True),    # level 0: too many pixels for the one-launch form
])
def test_groupnorm_split_equals_stats_then_split_rows(ops, B, P, C1, C2, act):
    """aldm_groupnorm_split (statistics + apply + SiLU + operand split in one launch up to 1024 pixels) writes the image of
    aldm_groupnorm_stats followed by aldm_split_rows BIT FOR BIT, and matches F.group_norm."""
    x1 = (torch.randn(B, P, C1, generator=g(1)) * 2 + 0.5).cuda()
    x2 = torch.randn(B, P, C2, generator=g(2)).cuda() if C2 else None
    ga = torch.randn(C1 + C2, generator=g(3)).cuda()
    be = torch.randn(C1 + C2, generator=g(4)).cuda()
    a = ops.ACT_SILU if act else ops.ACT_NONE
    s_new, r_new = ops.gn_split(x1, ga, be, groups=32, eps=1e-5, x2=x2, act=a, want_raw=True)
    sc, sh = ops.gn_stats(x1, ga, be, groups=32, eps=1e-5, x2=x2)
    s_old, r_old = ops.split_rows(x1, x2, pre=(sc, sh), act=a, want_raw=True)
    assert torch.equal(s_new.data, s_old.data) and torch.equal(r_new.data, r_old.data)
    xc = x1 if x2 is None else torch.cat([x1, x2], -1)
    ref = F.group_norm(xc.double().cpu().permute(0, 2, 1), 32, ga.double().cpu(), be.double().cpu(), 1e-5).permute(0, 2, 1)
    if act:
        ref = F.silu(ref)
    assert rel_err(s_new.float(), ref) < (5e-6 if exact_split(ops) else 3e-5)


@pytest.mark.parametrize("B,L,heads", [(16, 1024, 8), (4, 256, 12), (3, 64, 20), (2, 96, 2), (2, 32, 2)])
def test_qkv_epilogue_and_presplit_attention_are_bitwise_the_fp32_kv_path(ops, B, L, heads):
    """ALDM_EPI_QKV + aldm_attention_d32_presplit (k as a split image, v transposed per key tile straight from the accumulator
    layout) against the round-2 path (fp32 qkv, K / V split inside the attention kernel's key loop): the same products in the
    same order -> BIT-identical attention output; and both within the GEMM tolerance of fp64."""
    C = heads * 32
    x = torch.randn(B, L, C, generator=g(1))
    wq, wk, wv = (torch.randn(C, C, generator=g(2 + i)) / math.sqrt(C) for i in range(3))
    pw = ops.pack_conv(torch.cat([wq, wk, wv], 0))
    xs = ops.split_rows(x.cuda())
    qkv = ops.linear(xs, pw)
    a_old, s_old = ops.attention(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], heads, split_out="also")
    q, kimg, vtimg = ops.linear_qkv(xs, pw, heads, L)
    assert torch.equal(q, qkv[:, :, :C].contiguous())
    assert torch.equal(kimg.view(-1), ops.split_rows(qkv[:, :, C:2 * C].contiguous()).data.view(-1))
    a_new, s_new = ops.attention_presplit(q, kimg, vtimg, heads, split_out="also")
    assert torch.equal(a_new, a_old) and torch.equal(s_new.data, s_old.data)
    xd = xs.float().double().cpu()
    sh = lambda t: t.view(B, L, heads, 32).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sh(xd @ wq.double().t()), sh(xd @ wk.double().t()), sh(xd @ wv.double().t()))
    assert rel_err(a_new, ref.transpose(1, 2).reshape(B, L, C)) < fused_tol()


def _qkv_images(ops, q, k, v, heads):
    """(q, k image, v^T image) of GIVEN q / k / v [B, L, C] through the QKV epilogue, plus the fp32 [q | k | v] the same projection
    writes through the plain epilogue: the projection of [q | k | v] by the identity reproduces the values exactly in the 3-part
    form (x_hi + x_mid + x_lo against 1.0) and as x_hi + x_mid in the 2-part form — the images hold exactly the returned values."""
    B, L, C = q.shape
    xs = ops.split_rows(torch.cat([q, k, v], -1).cuda())
    pw = ops.pack_conv(torch.eye(3 * C))
    return ops.linear_qkv(xs, pw, heads, L), ops.linear(xs, pw)


def _score_tol(fp32_grade, smax):
    """What the rounding of a score of magnitude smax (natural units) costs the softmax, relative to max |out|: the 6-product form
    holds a dot product to ~2^-26 of its magnitude (measured 4e-9 per log2 unit of the largest score: 6.0e-6 at 1180), the
    3-product form with its 16-bit operands to ~2^-20 (measured 2.2e-7 .. 3.4e-7 per unit: 2.9e-4 at 930).  Bars <= 5x measured."""
    return (2.0 ** -26 if fp32_grade else 2.0 ** -20) * smax * 1.4426950408889634


def _late_large_scores(B, L, heads, jump):
    """q, k, v [B, L, C] with keys planted in LATER key tiles whose scores lie jump * (q.u) log2 units above (q.u > 0) or below
    (q.u < 0) everything else; a second, 1.5x larger key two tiles on.  The planted keys sit at in-tile indices 8 (held by the
    lower lane half of the S^T layout), 4 and 31 (upper lane half)."""
    C = heads * 32
    q = torch.randn(B, L, C, generator=g(1))
    k = torch.randn(B, L, C, generator=g(2))
    v = torch.randn(B, L, C, generator=g(3))
    u = torch.randn(heads, 32, generator=g(4))
    u = u / u.norm(dim=-1, keepdim=True)
    alpha = jump * math.sqrt(32.0) / 1.4426950408889634          # score of the planted key = jump * (q.u) in log2 units
    late = {32: [(20, 1.0)], 64: [(40, 1.0), (63, 1.0)], 128: [(40, 1.0), (100, 1.5), (127, 1.5)]}[L]
    if jump:
        for pos, f in late:
            k[:, pos, :] = (alpha * f * u).reshape(1, C)
    return q, k, v


@pytest.mark.parametrize("jump", [0.0, 12.0, 45.0, 70.0, 200.0])
@pytest.mark.parametrize("B,L,heads", [(16, 128, 8), (2, 128, 2), (1, 64, 2), (1, 32, 2)])
def test_attention_survives_late_large_scores(ops, B, L, heads, jump):
    """Scores hundreds of log2 units above everything seen so far, arriving in a later key tile — and in EITHER lane half of the
    S^T layout.  Until round 5 the pipelined kernels' cross-half exchange of the row maximum returned the LOWER half's value only
    (a compiler quirk around __builtin_bit_cast of a vector element, csrc/attn.hip max_across_halves): the "running maximum" was
    the maximum over half of each tile's keys — invisible while every key lay within 2^128 of it, non-finite output beyond (jump >=
    70 here).  Both paths — fp32 K / V, and the pre-split images — against the fp64 softmax, to the fused tolerance plus the
    rounding of the scores themselves, which grows with their magnitude (_score_tol), and bit-identical to each other.
    (16, 128, 8) runs 64 queries per wave, the others 32."""
    C = heads * 32
    q, k, v = _late_large_scores(B, L, heads, jump)
    (qi, kimg, vtimg), qkv = _qkv_images(ops, q, k, v, heads)
    assert torch.equal(qi, qkv[..., :C].contiguous())
    if exact_split(ops):
        assert torch.equal(qkv.cpu(), torch.cat([q, k, v], -1))
    a_new = ops.attention_presplit(qi, kimg, vtimg, heads)
    a_old = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
    sh = lambda t: t.double().cpu().view(B, L, heads, 32).transpose(1, 2)
    qd, kd, vd = (sh(qkv[..., i * C:(i + 1) * C].contiguous()) for i in range(3))
    sc = qd @ kd.transpose(-1, -2) / math.sqrt(32.0)
    ref = (torch.softmax(sc, -1) @ vd).transpose(1, 2).reshape(B, L, C)
    assert torch.isfinite(a_new).all() and torch.isfinite(a_old).all()
    assert torch.equal(a_new, a_old)
    assert rel_err(a_new, ref) < fused_tol() + _score_tol(exact_split(ops), float(sc.abs().max()))


@pytest.mark.parametrize("idx", [0, 4, 8, 15, 16, 23, 27, 31])
@pytest.mark.parametrize("tile", [0, 3])
def test_attention_row_maximum_covers_every_key_position(ops, tile, idx):
    """One key 200 log2 units above the rest at in-tile index `idx` of the first / last key tile (indices 4-7, 12-15, 20-23, 28-31
    are held by the upper lane half): the output is that key's value row, from both kernels."""
    B, L, heads = 1, 128, 2
    C = heads * 32
    q = torch.zeros(B, L, C)
    q[..., 0] = 1.0
    q[..., 32] = 1.0
    k = torch.zeros(B, L, C)
    k[0, 32 * tile + idx, 0] = k[0, 32 * tile + idx, 32] = 200.0 * math.sqrt(32.0) / 1.4426950408889634
    v = torch.randn(B, L, C, generator=g(5))
    (qi, kimg, vtimg), qkv = _qkv_images(ops, q, k, v, heads)
    a_new = ops.attention_presplit(qi, kimg, vtimg, heads)
    a_old = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
    want = qkv[0, 32 * tile + idx, 2 * C:].expand(L, C)
    assert torch.equal(a_new[0], want) and torch.equal(a_old[0], want)


@pytest.mark.parametrize("sched", ["0", "2"])
def test_presplit_kernels_selected_by_env_agree_with_the_default(sched):
    """ALDM_ATTN_SCHED is read once per process: a fresh one for 0 (the round-3 / 4 pipelined kernel: BIT-identical to the fp32-K/V
    path, like the default) and 2 (the opt-in one-pass loop with a fixed softmax reference per row: its probabilities differ from
    the exact-max kernels' by one common factor per row that cancels in O / l — equal to fp32 rounding, measured 4e-7 .. 1.8e-6
    of max |out| in the 6-product mode and <= 1e-5 with 2-part operands, plus _score_tol under extreme scores; its slow path — reference raised when a tile's row sum
    passes 2^60, exponentials that overflowed recomputed from the intact scores — is what the large jumps exercise)."""
    import os
    import subprocess
    import sys
    code = r"""
import math, sys, torch
sys.path.insert(0, "tests")
from test_dma_gpu import _late_large_scores, _qkv_images
from audioldm2_amd import ops
for (B, L, heads), jump in [((16, 1024, 8), 0.0), ((3, 64, 20), 0.0), ((2, 96, 2), 0.0), ((2, 32, 2), 0.0), ((16, 128, 8), 12.0),
                            ((16, 128, 8), 70.0), ((2, 128, 2), 200.0), ((1, 64, 2), 200.0), ((1, 32, 2), 200.0)]:
    C = heads * 32
    if L in (32, 64, 128):
        q, k, v = _late_large_scores(B, L, heads, jump)
    else:
        q, k, v = (torch.randn(B, L, C, generator=torch.Generator().manual_seed(i)) for i in (1, 2, 3))
    (qi, kimg, vtimg), qkv = _qkv_images(ops, q, k, v, heads)
    a_new = ops.attention_presplit(qi, kimg, vtimg, heads)
    a_old = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
    sh = lambda t: t.double().cpu().view(B, L, heads, 32).transpose(1, 2)
    qd, kd, vd = (sh(qkv[..., i * C:(i + 1) * C].contiguous()) for i in range(3))
    sc = qd @ kd.transpose(-1, -2) / math.sqrt(32.0)
    ref = (torch.softmax(sc, -1) @ vd).transpose(1, 2).reshape(B, L, C)
    rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
    print("CASE", B, L, heads, jump, int(torch.isfinite(a_new).all()), int(torch.equal(a_new, a_old)), rel(a_new, a_old), rel(a_new, ref),
          float(sc.abs().max()))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("bf16x6", "bf16x3"):
        env = dict(os.environ, ALDM_MMA=mode, ALDM_ATTN_SCHED=sched, PYTHONPATH=root)
        env.pop("ALDM_ATTN_MMA", None)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l.split()[1:] for l in out.stdout.splitlines() if l.startswith("CASE")]
        assert len(lines) == 9, out.stdout[-2000:]
        grade = 5e-6 if mode == "bf16x6" else 5e-5
        for B, L, heads, jump, finite, bitwise, d_old, d_ref, smax in lines:
            what = (mode, sched, B, L, heads, jump)
            score_tol = _score_tol(mode == "bf16x6", float(smax))
            assert finite == "1", what
            log_err(float(d_ref), grade + score_tol, f"attention ALDM_ATTN_SCHED={sched} {mode} vs fp64")
            assert float(d_ref) < grade + score_tol, what
            if sched == "0":
                assert bitwise == "1", what
            else:
                log_err(float(d_old), (3e-6 if mode == "bf16x6" else 3e-5) + score_tol, f"attention ALDM_ATTN_SCHED=2 {mode} vs the exact-max kernel")
                assert float(d_old) < (3e-6 if mode == "bf16x6" else 3e-5) + score_tol, what


# ---- the operand-stationary form for short K (csrc/igemm_dma_os.h): aldm_igemm_force(32, 128, ..., 300 + ring depth) ------------
# Same products in the same order per k-tile as igemm_dma_kernel, but accumulated by v_mfma_f32_16x16x32_bf16 (32 k per
# instruction) instead of two chained 32x32x16 ones: equal to fp32 rounding, not bitwise.
OS_TOL = 2e-6


def _os_depths(ops, K):
    if ops.split_parts() == 3:
        return [2, 3] if K == 256 else [2]
    return [2, 3, 4] if K == 256 else [2, 3]


def _os_vs_classic(ops, st, fn):
    """fn() under the operand-stationary kernel and under igemm_dma_kernel (64x128 tile)."""
    ops.igemm_force(32, 128, 1, 0, 300 + st)
    try:
        y_os = fn()
    finally:
        ops.igemm_force(0, 0, 0)
    ops.igemm_force(64, 128, 1, 0, 2)
    try:
        y_old = fn()
    finally:
        ops.igemm_force(0, 0, 0)
    return y_os, y_old


def _img_tol(ops):
    """Split images of two fp32 tensors that agree to OS_TOL: exact 3-part images agree as well; in a 2-part image a value on a
    rounding boundary of `mid` may land on the other side (2^-17 relative)."""
    return OS_TOL if exact_split(ops) else OS_TOL + 2.0 ** -16


def _parts_to_float(img, part_dim):
    """Raw int16 part images (k / v^T of ALDM_EPI_QKV) -> fp32 values (sum of the parts)."""
    return (img.to(torch.int32) << 16).view(torch.float32).sum(dim=part_dim)


@pytest.mark.parametrize("K,N,M", [(256, 256, 16384), (256, 768, 16384), (384, 384, 4096), (256, 128, 32 * 37 + 5),
                                   (256, 640, 100), (384, 1152, 4096 + 17)])
def test_dma_os_linear_matches_classic(ops, K, N, M):
    """Plain / bias / residual / alpha / both outputs: the UNet's K = C projections at the BASELINE batch (16384 x 256 -> 256 / 768,
    4096 x 384 -> 384), ragged M (rows past the end are fetched from the last row and discarded; the last block's chunk is
    short), N that is not a whole number of 128-column slabs, more and fewer blocks than compute units."""
    x = torch.randn(1, M, K, generator=g(1))
    w = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    b = torch.randn(N, generator=g(3))
    res = torch.randn(1, M, N, generator=g(4)).cuda()
    pw = ops.pack_conv(w, b)
    pw0 = ops.pack_conv(w, None)
    xs = ops.split_rows(x.cuda())
    ref = xs.float().double().cpu() @ w.double().t()
    for st in _os_depths(ops, K):
        y_os, y_old = _os_vs_classic(ops, st, lambda: ops.linear(xs, pw0))
        assert rel_err(y_os, ref) < gemm_tol(), st
        assert rel_err(y_os, y_old) < OS_TOL, st
        (y_os, s_os), (y_old, s_old) = _os_vs_classic(ops, st, lambda: ops.linear(xs, pw, res=res, alpha=0.5, split_out="also"))
        assert rel_err(y_os, 0.5 * (ref + b.double() + res.double().cpu())) < gemm_tol(), st
        assert rel_err(y_os, y_old) < OS_TOL and rel_err(s_os.float(), s_old.float()) < _img_tol(ops), st
        assert_split_equals(ops, s_os, y_os, st)
        s_only, _ = _os_vs_classic(ops, st, lambda: ops.linear(xs, pw, res=res, alpha=0.5, split_out="only"))
        assert torch.equal(s_only.data, s_os.data), st     # the same kernel twice: deterministic


@pytest.mark.parametrize("C,M", [(256, 16384), (384, 4096), (256, 32 * 9 + 7)])
def test_dma_os_geglu_matches_classic(ops, C, M):
    """The GEGLU epilogue: a wave's 16 columns are 8 value columns and their 8 gate columns of the packed weight image; value *
    gelu(gate) is evaluated in registers (one DPP row rotation) with the classic kernel's arithmetic (attention.py:37-45)."""
    x = torch.randn(1, M, C, generator=g(1))
    w1 = torch.randn(8 * C, C, generator=g(4)) / math.sqrt(C)
    b1 = torch.randn(8 * C, generator=g(5))
    pw = ops.pack_geglu(w1, b1)
    xs = ops.split_rows(x.cuda())
    h = xs.float().double().cpu() @ w1.double().t() + b1.double()
    a, gate = h.chunk(2, -1)
    ref = a * F.gelu(gate)
    for st in _os_depths(ops, C):
        (y_os, s_os), (y_old, s_old) = _os_vs_classic(ops, st, lambda: ops.linear_geglu(xs, pw, split_out="also"))
        assert rel_err(y_os, ref) < fused_tol(), st
        assert rel_err(y_os, y_old) < OS_TOL and rel_err(s_os.float(), s_old.float()) < _img_tol(ops), st
        assert_split_equals(ops, s_os, y_os, st)
        s_only, _ = _os_vs_classic(ops, st, lambda: ops.linear_geglu(xs, pw, split_out="only"))
        assert torch.equal(s_only.data, s_os.data), st


@pytest.mark.parametrize("B,L,heads", [(16, 1024, 8), (4, 256, 12), (2, 96, 8)])
def test_dma_os_qkv_epilogue_matches_classic(ops, B, L, heads):
    """ALDM_EPI_QKV from the operand-stationary kernel: q fp32, k as a split image, v transposed per key tile — the classic
    kernel's three outputs to fp32 rounding (C = 256 / 384: whole 128-column slabs per segment), and the attention over them
    within the GEMM tolerance of fp64."""
    C = heads * 32
    x = torch.randn(B, L, C, generator=g(1))
    wq, wk, wv = (torch.randn(C, C, generator=g(2 + i)) / math.sqrt(C) for i in range(3))
    pw = ops.pack_conv(torch.cat([wq, wk, wv], 0))
    xs = ops.split_rows(x.cuda())
    for st in _os_depths(ops, C):
        (q1, k1, v1), (q0, k0, v0) = _os_vs_classic(ops, st, lambda: ops.linear_qkv(xs, pw, heads, L))
        assert rel_err(q1, q0) < OS_TOL, st
        assert rel_err(_parts_to_float(k1, 2), _parts_to_float(k0, 2)) < _img_tol(ops), st
        assert rel_err(_parts_to_float(v1, 3), _parts_to_float(v0, 3)) < _img_tol(ops), st
        a = ops.attention_presplit(q1, k1, v1, heads)
        xd = xs.float().double().cpu()
        sh = lambda t: t.view(B, L, heads, 32).transpose(1, 2)
        ref = F.scaled_dot_product_attention(sh(xd @ wq.double().t()), sh(xd @ wk.double().t()), sh(xd @ wv.double().t()))
        assert rel_err(a, ref.transpose(1, 2).reshape(B, L, C)) < fused_tol(), st


def test_dma_os_refuses_what_it_cannot_run_and_hints_fall_back(ops):
    """A FORCED operand-stationary launch outside its domain (K = 128: no instantiation) fails loudly; the same request as a tuned
    HINT (tables are keyed by geometry) falls back to aldm_igemm's own tile choice and still computes the right thing."""
    x = torch.randn(1, 8, 8, 128, generator=g(1))
    w = torch.randn(128, 128, generator=g(2)) / math.sqrt(128)
    pw = ops.pack_conv(w, None)
    xs = ops.split_rows(x.cuda())
    ops.igemm_force(32, 128, 1, 0, 302)
    try:
        with pytest.raises(RuntimeError):
            ops.linear(xs, pw)
    finally:
        ops.igemm_force(0, 0, 0)
    tab = ops._tuned_table("dma2" if ops.split_parts() == 2 else "dma")
    d_keys = []
    ops.TUNE_LOG = d_keys
    try:
        y0 = ops.linear(xs, pw)
    finally:
        ops.TUNE_LOG = None
    had = tab.get(d_keys[0])
    tab[d_keys[0]] = [32, 128, 1, 302]
    try:
        y1 = ops.linear(xs, pw)
    finally:
        if had is None:
            del tab[d_keys[0]]
        else:
            tab[d_keys[0]] = had
    assert rel_err(y0, y1) < OS_TOL
    assert rel_err(y1, xs.float().double().cpu() @ w.double().t()) < gemm_tol()


# ---- the halo-patch 3x3 convolution kernel (csrc/igemm_dma_halo.h) -------------------------------------------
# (tile, stages code 400 + 10 * w8 + weight-ring depth) per mode: every instantiation of igemm_dma_halo.hip
_HALO = {"bf16x6": [(256, 128, 402), (128, 128, 403), (128, 128, 404), (128, 128, 402), (128, 128, 412), (128, 128, 413)],
         "bf16x3": [(256, 128, 402), (256, 128, 403), (128, 128, 403), (128, 128, 404), (128, 128, 413)]}


def _halo_case(ops, bm, bn, st, B, C, N, H, W, splits=1, epilogue=True):
    """conv3x3(x) (+ bias, timestep row bias, SiLU, residual) on the halo kernel against fp64 and against igemm_dma_kernel."""
    x = torch.randn(B, C, H, W, generator=g(1))
    w = torch.randn(N, C, 3, 3, generator=g(2)) / math.sqrt(C * 9)
    b = torch.randn(N, generator=g(3))
    emb = torch.randn(B, N, generator=g(4))
    res = torch.randn(B, N, H, W, generator=g(5))
    conv = F.conv2d(x, w, b, padding=1)
    ref = F.silu(conv + emb[:, :, None, None]) + res if epilogue else conv
    pw = ops.pack_conv(w, b)
    xs = ops.split_rows(cl(x))
    kw = dict(rowbias=emb.cuda(), act=ops.ACT_SILU, res=cl(res)) if epilogue else {}
    y_classic = ops.conv(xs, pw, pad=(1, 1), **kw)
    ops.igemm_force(bm, bn, splits, 0, st)
    try:
        y, s = ops.conv(xs, pw, pad=(1, 1), split_out="also", **kw)
        y2 = ops.conv(xs, pw, pad=(1, 1), **kw)
    finally:
        ops.igemm_force(0, 0, 0)
    tol = fused_tol() if epilogue else gemm_tol()
    assert rel_err(uncl(y), ref) < tol
    assert torch.equal(y, y2), "must be bitwise reproducible"
    assert_split_equals(ops, s, y)
    # the same products in another summation order (channel block outer, tap inner): fp32 rounding apart
    assert rel_err(y, y_classic) < (2e-6 if exact_split(ops) else 1e-5)


@pytest.mark.parametrize("mode,bm,bn,st", [(m, *t) for m, ts in _HALO.items() for t in ts])
def test_dma_halo_every_instantiation(mode, bm, bn, st):
    """Every instantiation on the UNet's level-0 geometry in small (W = 16, two images, 128 -> 192 channels: a ragged second
    column tile), full epilogue, split-image second output.  (The 4-deep weight ring of the 4-wave 128-row tile leaves 7 patch-piece
    slots per channel block: 28 pieces, enough for W <= 8 with 3-part images — that instantiation runs the level-1 geometry.)"""
    from audioldm2_amd import ops
    prev = ops.set_mma(mode)
    try:
        if st == 404 and mode == "bf16x6":
            _halo_case(ops, bm, bn, st, B=2, C=128, N=192, H=64, W=8)
        else:
            _halo_case(ops, bm, bn, st, B=2, C=128, N=192, H=32, W=16)
    finally:
        ops.set_mma(prev)


@pytest.mark.parametrize("W,H,bm,st", [(16, 256, 256, 402), (8, 128, 256, 402), (8, 128, 128, 403), (4, 64, 256, 402), (4, 64, 128, 403),
                                       (2, 64, 128, 403), (32, 16, 128, 412), (64, 8, 128, 412), (16, 16, 256, 402), (16, 8, 128, 403)])
def test_dma_halo_image_widths(ops, W, H, bm, st):
    """Every image width the sampling path has (UNet levels 16 / 8 / 4 / 2, VAE decoder 16 / 32 / 64), tiles at the top and bottom
    image border and in the middle, a tile that is a whole image (zero rows on both sides), a half-used last patch chunk (W = 4)."""
    if ops.split_parts() == 2 and st == 402 and bm == 128:
        st = 403
    if ops.split_parts() == 2 and st == 412:
        st = 413
    _halo_case(ops, bm, 128, st, B=2, C=64, N=128, H=H, W=W, epilogue=False)


@pytest.mark.parametrize("splits", [2, 3])
def test_dma_halo_split_k_cuts_between_channel_blocks(ops, splits):
    """Split-K of the halo kernel hands whole 32-channel blocks (9 k-tiles each) to every split: C = 192 = 6 blocks -> 3 + 3 /
    2 + 2 + 2; the workspace reduce applies the epilogue."""
    _halo_case(ops, 128, 128, 403, B=1, C=192, N=128, H=32, W=8, splits=splits)


def test_dma_halo_refuses_what_it_cannot_run_and_hints_fall_back(ops):
    """FORCED onto a launch outside its domain (stride 2; a 1x1 conv; an image whose rows do not tile) the halo kernel fails
    loudly; as a tuned HINT (tables are keyed by geometry) the same request falls back to aldm_igemm's own choice."""
    x = torch.randn(1, 64, 24, 16, generator=g(1))          # OH * OW = 384: not a multiple of the 256-row tile
    w = torch.randn(64, 64, 3, 3, generator=g(2)) / math.sqrt(64 * 9)
    pw = ops.pack_conv(w, None)
    xs = ops.split_rows(cl(x))
    ops.igemm_force(256, 128, 1, 0, 402)
    try:
        with pytest.raises(RuntimeError, match="halo-patch"):
            ops.conv(xs, pw, pad=(1, 1))
        with pytest.raises(RuntimeError, match="halo-patch"):
            ops.conv(xs, pw, stride=(2, 2), pad=(1, 1))
    finally:
        ops.igemm_force(0, 0, 0)
    tab = ops._tuned_table("dma2" if ops.split_parts() == 2 else "dma")
    keys = []
    ops.TUNE_LOG = keys
    try:
        y0 = ops.conv(xs, pw, pad=(1, 1))
    finally:
        ops.TUNE_LOG = None
    had = tab.get(keys[0])
    tab[keys[0]] = [256, 128, 1, 402]
    try:
        y1 = ops.conv(xs, pw, pad=(1, 1))
    finally:
        if had is None:
            del tab[keys[0]]
        else:
            tab[keys[0]] = had
    assert torch.equal(y0, y1) or rel_err(y0, y1) < 2e-6
    assert rel_err(uncl(y1), F.conv2d(x, w, padding=1)) < gemm_tol()


@pytest.mark.parametrize("B,L,heads", [(16, 1024, 8), (16, 256, 12), (2, 128, 2), (2, 512, 4), (1, 256, 2), (3, 64, 20), (16, 128, 8),
                                       (2, 384, 2)])
def test_presplit_attention_with_kv_through_lds_is_bitwise_the_default(ops, B, L, heads):
    """aldm_attention_sched(3): the K / V^T tiles of a (sample, head) cross L2 -> LDS once per block instead of once per wave (an
    NST-deep ring of LDS-DMA pieces, one s_barrier per key tile); the arithmetic and its order are the default kernel's, so the
    outputs are BIT-identical — at one, two, three and many key tiles, 32 and 64 queries per wave, and on launches with partial
    blocks (64 / 128 / 384 queries: those stay on the default kernel).  In one process, through the setter."""
    C = heads * 32
    q, k, v = (torch.randn(B, L, C, generator=g(i)) for i in (1, 2, 3))
    (qi, kimg, vtimg), qkv = _qkv_images(ops, q, k, v, heads)
    prev = ops.attention_sched(1)
    try:
        a1 = ops.attention_presplit(qi, kimg, vtimg, heads)
        assert ops.attention_sched(3) == 1
        a3 = ops.attention_presplit(qi, kimg, vtimg, heads)
        a3b, s3 = ops.attention_presplit(qi, kimg, vtimg, heads, split_out="also")
    finally:
        ops.attention_sched(prev)
    assert torch.equal(a1, a3) and torch.equal(a3, a3b)
    assert_split_equals(ops, s3, a3)
    sh = lambda t: t.double().cpu().view(B, L, heads, 32).transpose(1, 2)
    qd, kd, vd = (sh(qkv[..., i * C:(i + 1) * C].contiguous()) for i in range(3))
    ref = F.scaled_dot_product_attention(qd, kd, vd).transpose(1, 2).reshape(B, L, C)
    assert rel_err(a3, ref) < fused_tol()
