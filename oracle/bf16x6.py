"""oracle/bf16x6.py — numpy restatement of the bf16-split ("BF16x6") arithmetic and data format of the igemm engine.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

This piece has no counterpart in the reference (which multiplies in fp32 through ATen): it is the checker for HOW the
HIP kernels evaluate the reference's fp32 products on the bf16 matrix cores (audioldm2_amd/csrc/igemm_kernel.h,
docs/experiments_r1-r6.md §3.1b), i.e. that the evaluation is fp32-grade and that the weight image has the documented layout.
  * split3        x = hi + mid + lo exactly, each part the top 16 bits of an fp32 (truncation)
  * product6      a*b ~ hi*lo + lo*hi + mid*mid + hi*mid + mid*hi + hi*hi (exact partial products, here summed in
                  fp64; the MFMA accumulates them in fp32, which is the fp32 kernels' accumulation error too)
  * split_image   packed weight [ceil(K/4)][Npad][4] -> [4*ceil(K/32)][3][Npad][8] uint16 (aldm_pack_split_bf16)

"bf16x3" (the DMA-fed GEMMs' default since round 2, audioldm2_amd/csrc/igemm_epilogue.h split4_rn2, igemm_dma.h NP = 2):
  * split2_rn     hi = RN_bf16(x), mid = RN_bf16(x - hi) (round to nearest even): |x - hi - mid| <= 2^-16 |x| (2^-18 rms), unbiased
  * matmul3       a*b ~ mid*hi + hi*mid + hi*hi
  * split_image(parts=2)  [4*ceil(K/32)][2][Npad][8] uint16 (aldm_pack_split_bf16_parts)
"""
import numpy as np

_MASK = np.uint32(0xFFFF0000)


def _bf16_rn(x: np.ndarray) -> np.ndarray:
    """fp32 -> nearest bf16 (ties to even), returned as fp32 with zero low half (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & _MASK
    return r.view(np.float32)


def split2_rn(x: np.ndarray):
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi = _bf16_rn(x)
    mid = _bf16_rn(x - hi)          # x - hi is exact (|x - hi| <= 2^-9 |x|, same binade or below)
    return hi, mid


def split2_bits(x: np.ndarray):
    return tuple((p.view(np.uint32) >> np.uint32(16)).astype(np.uint16) for p in split2_rn(x))


def matmul3(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """a [M, K] @ b [K, N] with every scalar product evaluated as the three bf16x3 partial products (ideal accumulation)."""
    ah, am = (p.astype(np.float64) for p in split2_rn(a))
    bh, bm = (p.astype(np.float64) for p in split2_rn(b))
    return am @ bh + ah @ bm + ah @ bh


def split3(x: np.ndarray):
    """fp32 array -> (hi, mid, lo) fp32 arrays whose low 16 bits are zero and whose sum is x exactly."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi = (x.view(np.uint32) & _MASK).view(np.float32)
    r1 = x - hi                                   # exact: hi shares x's sign and leading bits
    mid = (r1.view(np.uint32) & _MASK).view(np.float32)
    r2 = r1 - mid                                 # exact
    lo = (r2.view(np.uint32) & _MASK).view(np.float32)
    return hi, mid, lo


def split3_bits(x: np.ndarray):
    """The three parts as bf16 bit patterns (uint16), as stored in LDS / in the weight image."""
    return tuple((p.view(np.uint32) >> np.uint32(16)).astype(np.uint16) for p in split3(x))


def matmul6(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """a [M, K] @ b [K, N] with every scalar product evaluated as the six bf16 partial products (ideal accumulation)."""
    ah, am, al = (p.astype(np.float64) for p in split3(a))
    bh, bm, bl = (p.astype(np.float64) for p in split3(b))
    return ah @ bl + al @ bh + am @ bm + ah @ bm + am @ bh + ah @ bh


def pack_kn(w_kn: np.ndarray) -> np.ndarray:
    """[K, N] -> the igemm packed layout [ceil(K/4)][Npad][4] (aldm_pack_weight / aldm_pack_kn), zero padded."""
    K, N = w_kn.shape
    Kg, Npad = (K + 3) // 4, (N + 31) // 32 * 32
    out = np.zeros((Kg * 4, Npad), np.float32)
    out[:K, :N] = w_kn
    return np.ascontiguousarray(out.reshape(Kg, 4, Npad).transpose(0, 2, 1))


def split_image(packed: np.ndarray, K: int, parts: int = 3) -> np.ndarray:
    """packed [Kg][Npad][4] fp32 -> [Ko][parts][Npad][8] uint16 with Ko = 4*ceil(K/32) k-octets (zero padded to whole
    k-tiles); element j of a slot is k = 8*ko + j, part 0/1/2 = hi/mid/lo (parts = 2: hi/mid, rounded to nearest)."""
    Kg, Npad, _ = packed.shape
    Ko = 4 * ((K + 31) // 32)
    full = np.zeros((Ko * 2, Npad, 4), np.float32)
    full[:Kg] = packed
    octets = full.reshape(Ko, 2, Npad, 4).transpose(0, 2, 1, 3).reshape(Ko, Npad, 8)  # [ko][n][j]
    bits = split3_bits(octets) if parts == 3 else split2_bits(octets)
    return np.ascontiguousarray(np.stack(bits, axis=1))  # [ko][part][n][j]
