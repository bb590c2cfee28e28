"""CPU checks of the bf16-split arithmetic the HIP igemm kernels use by default (oracle/bf16x6.py, docs/experiments_r1-r6.md §3.1b):
the operand split is exact, six partial products reproduce an fp32 product to <= 2^-21 relative in the worst case and
to about one fp32 rounding on average, and a whole
contraction evaluated that way is at least as close to the exact result as an fp32 GEMM."""
import numpy as np

from oracle import bf16x6 as bx


def _samples(rng, n):
    x = rng.standard_normal(n).astype(np.float32) * np.exp(rng.uniform(-20, 20, n)).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 3.0, 1.0 + 2.0**-23, 65504.0, 3.4e38, -3.4e38, 1.17549435e-38, 255.0,
                     256.0, 0.1, -1e-30], np.float32)
    return np.concatenate([x, edge])


def test_split_is_exact_and_parts_are_bf16():
    x = _samples(np.random.default_rng(0), 20000)
    hi, mid, lo = bx.split3(x)
    for p in (hi, mid, lo):
        assert not np.any(p.view(np.uint32) & np.uint32(0xFFFF))  # bf16 bit patterns
    # hi + mid needs <= 16 significant bits, + lo <= 24: both additions are exact in fp32
    assert np.array_equal((hi + mid) + lo, x)
    nz = x != 0
    assert np.all(np.abs(mid[nz]) <= np.abs(hi[nz]) * 2.0**-7) and np.all(np.abs(lo[nz]) <= np.abs(hi[nz]) * 2.0**-15)
    assert np.all((np.sign(mid) == np.sign(x)) | (mid == 0)) and np.all((np.sign(lo) == np.sign(x)) | (lo == 0))


def test_six_partial_products_are_fp32_grade():
    rng = np.random.default_rng(1)
    a, b = _samples(rng, 20000)[:20000], _samples(rng, 20000)[:20000]
    a = np.clip(a, -1e18, 1e18)
    b = np.clip(b, -1e18, 1e18)
    (ah, am, al), (bh, bm, bl) = bx.split3(a), bx.split3(b)
    f = np.float64
    six = f(ah) * f(bl) + f(al) * f(bh) + f(am) * f(bm) + f(ah) * f(bm) + f(am) * f(bh) + f(ah) * f(bh)
    exact = f(a) * f(b)
    nz = exact != 0
    rel = np.abs(six[nz] - exact[nz]) / np.abs(exact[nz])
    # dropped terms: mid*lo + lo*mid + lo*lo, with |mid| < 2^-7 |hi| and |lo| < 2^-15 |hi| (truncation splits)
    assert rel.max() <= 2.0**-21
    # on average the error is that of one fp32 rounding (which is <= 2^-24, 0.25..0.5 * 2^-24 typically)
    assert rel.mean() <= 2.0**-24
    rn = np.abs(np.float64(np.float32(exact[nz])) - exact[nz]) / np.abs(exact[nz])
    assert rel.mean() <= 4 * rn.mean()


def test_contraction_error_is_no_worse_than_an_fp32_gemm():
    rng = np.random.default_rng(2)
    M, K, N = 128, 1152, 96
    a = (rng.standard_normal((M, K)) * np.exp(rng.standard_normal((M, K)))).astype(np.float32)
    b = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    e6 = np.abs(bx.matmul6(a, b) - ref).max() / np.abs(ref).max()
    e32 = np.abs((a @ b).astype(np.float64) - ref).max() / np.abs(ref).max()
    assert e6 < 2e-7 and e6 <= e32


def test_split_image_layout():
    rng = np.random.default_rng(3)
    K, N = 72, 40  # ragged: 2.25 k-tiles, Npad = 64
    w = rng.standard_normal((K, N)).astype(np.float32)
    packed = bx.pack_kn(w)
    assert packed.shape == (18, 64, 4) and packed[5, 7, 2] == w[22, 7]
    img = bx.split_image(packed, K)
    assert img.shape == (12, 3, 64, 8) and img.dtype == np.uint16
    hi, mid, lo = bx.split3_bits(w)
    ko, j, n = 4, 3, 17  # k = 35
    assert (img[ko, 0, n, j], img[ko, 1, n, j], img[ko, 2, n, j]) == (hi[35, n], mid[35, n], lo[35, n])
    assert not img[9:].any() and not img[:, :, 40:].any()  # k >= 72 and n >= 40 are zero padding
    parts = [(img[:, q].astype(np.uint32) << np.uint32(16)).view(np.float32) for q in range(3)]
    back = ((parts[0] + parts[1]) + parts[2]).transpose(0, 2, 1).reshape(12 * 8, 64)  # [ko][n][j] -> [k][n]
    assert np.array_equal(back[:K, :N], w)


# ---- "bf16x3": (hi, mid) rounded to nearest, three partial products --------------------------------------------------
def test_two_part_split_is_16_bit_and_unbiased():
    x = _samples(np.random.default_rng(4), 20000)
    x = x[np.abs(x) < 1e38]
    hi, mid = bx.split2_rn(x)
    for p in (hi, mid):
        assert not np.any(p.view(np.uint32) & np.uint32(0xFFFF))
    nz = x != 0
    err = (np.float64(hi) + np.float64(mid) - np.float64(x))[nz] / np.float64(x)[nz]
    assert np.abs(err).max() <= 2.0**-16          # two nearest roundings of 8 significant bits each; ~2^-18 rms
    assert abs(err.mean()) <= 2.0**-24            # no systematic shrink (a truncation split would sit at -2^-17)
    assert np.abs(err).mean() <= 2.0**-19


def test_three_partial_products_error_budget():
    """Per product: operand roundings (2 x <= 2^-16) + the dropped mid*mid (<= 2^-16): <= 3 * 2^-16 worst case, ~2^-18 rms,
    unbiased; a whole contraction lands at ~4.4e-6 relative rms — 20x an fp32 GEMM, 500x better than plain bf16."""
    rng = np.random.default_rng(5)
    a = rng.standard_normal(20000).astype(np.float32)
    b = rng.standard_normal(20000).astype(np.float32)
    (ah, am), (bh, bm) = bx.split2_rn(a), bx.split2_rn(b)
    f = np.float64
    three = f(am) * f(bh) + f(ah) * f(bm) + f(ah) * f(bh)
    exact = f(a) * f(b)
    rel = (three - exact) / exact
    assert np.abs(rel).max() <= 3 * 2.0**-16 and np.sqrt((rel ** 2).mean()) <= 2.0**-17.5 and abs(rel.mean()) <= 2.0**-21
    M, K, N = 128, 1152, 96
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    ref = A.astype(f) @ B.astype(f)
    rms = lambda e: np.sqrt((e ** 2).mean())
    e3 = rms(bx.matmul3(A, B) - ref) / rms(ref)
    bf = lambda t: bx._bf16_rn(t).astype(f)
    e1 = rms(bf(A) @ bf(B) - ref) / rms(ref)
    assert e3 < 8e-6 and e1 > 100 * e3


def test_two_part_split_image_layout():
    rng = np.random.default_rng(6)
    K, N = 72, 40
    w = rng.standard_normal((K, N)).astype(np.float32)
    packed = bx.pack_kn(w)
    img = bx.split_image(packed, K, parts=2)
    assert img.shape == (12, 2, 64, 8) and img.dtype == np.uint16
    hi, mid = bx.split2_bits(w)
    assert img[1, 0, 5, 3] == hi[11, 5] and img[1, 1, 5, 3] == mid[11, 5]   # k = 8*1 + 3
    assert not img[9:].any() and not img[:, :, 40:].any()                    # zero padding
