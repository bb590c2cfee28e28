"""Thin tensor-level wrappers over the C ABI (lib.py).  PyTorch is plumbing here: device memory,
the current HIP stream and nothing else — every FLOP of the hot path runs in libaldm_hip.so.

Layout convention: activations are channels-last fp32, `[B, H, W, C]` (2-D) or `[B, 1, L, C]` (1-D).
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import lib as _l
from .lib import (ACT_GELU, ACT_GELU_TANH, ACT_LOGCLAMP, ACT_LRELU, ACT_NONE, ACT_SILU, ACT_TANH, B_NT, B_PACKED,
                  IgemmDesc)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError(f"{name}: expected a contiguous fp32 CUDA tensor, got "
                           f"{t.dtype} {t.device} contiguous={t.is_contiguous()}")


# Matrix-core path of the igemm engine ($ALDM_MMA overrides; set_mma() switches at run time; packed weights build their
# split images lazily on first use).  Operands, accumulation and every stored tensor are fp32 in all three:
#   "f32"     fp32 MFMA (v_mfma_f32_32x32x2_f32): exact fp32 products, 1/16 of the bf16 matrix rate;
#   "bf16x6"  the DEFAULT since round 4: every fp32 product as 6 bf16 partial products of exact 3-way operand splits —
#             fp32-grade (error vs fp64 2.4e-7 rms per contraction, the fp32 MFMA's 2.1e-7), i.e. not narrower than the
#             reference's fp32 multiply (openaimodel.py:492,531 `use_fp16=False`).  It is the mode the GPU suite runs and the
#             mode bench.py's headline is measured in;
#   "bf16x3"  the opt-in FAST mode (the default of rounds 2-3): the DMA-fed GEMMs (pre-split operands, csrc/igemm_dma.h) and the
#             attention keep two parts per operand, (hi, mid) rounded to nearest = 16 significant bits, and evaluate
#             hi*hi + hi*mid + mid*hi — half the matrix-pipe work and 2/3 of the operand bytes.  Per-GEMM error 4.4e-6 rms vs
#             fp64 (plain bf16: 2.9e-3; tools/x3_accuracy.py); the 200-step waveform matches the reference to 1.4e-6 rms, 700x
#             inside north_star's 1e-3 — but its operands ARE narrower than fp32, so it is reported as a named sub-record
#             (`fast`), never as the headline.  Launches without a pre-split operand run "bf16x6" in this mode too.
#   "f16x3"   round 6, opt-in, reported as a named sub-record: the GEMMs whose A operand comes out of a GroupNorm or a LayerNorm — the
#             ResBlocks' 3x3 convs, proj_in, q / k / v, the GEGLU projection; the VAE's ResnetBlock convs: ~2/3 of the UNet's FLOPs —
#             take 2-part IEEE-fp16 images of power-of-two scaled operands (a normalised tensor has an a-priori bound, sqrt(n) max|gamma|
#             + max|beta|, so the scale is known before the data; weights are scaled by their maximum) and evaluate hi*hi + hi*lo +
#             lo*hi on the fp16 matrix instruction: THREE MFMAs per fp32 product with 22 of 24 significand bits per operand — measured
#             0.65-0.7x the fp32 MFMA's own error against fp64 (profiles/r06_f16x3_accuracy.txt).  Every other launch (operands
#             without a bound: GEMM / attention outputs, raw activations) runs "bf16x6", attention included.
MMA_MODE = os.environ.get("ALDM_MMA", "bf16x6")
MMA_MODES = ("f32", "bf16x6", "bf16x3", "f16x3")
assert MMA_MODE in MMA_MODES, MMA_MODE


def set_mma(mode: str) -> str:
    """Select the matrix-core path for subsequent igemm launches ("f32" | "bf16x6" | "bf16x3" | "f16x3"); returns the previous
    one.  Captured HIP graphs keep the path they were captured with."""
    global MMA_MODE
    assert mode in MMA_MODES, mode
    prev, MMA_MODE = MMA_MODE, mode
    if _l._lib is not None or os.path.exists(_l.LIB_PATH):
        try:  # the attention kernel follows the engine's mode unless $ALDM_ATTN_MMA pins it
            if "ALDM_ATTN_MMA" not in os.environ:
                _l.load().aldm_attention_mma({"f32": 1, "bf16x6": 2, "bf16x3": 3, "f16x3": 2}[mode])
        except RuntimeError:
            pass
    return prev


def split_parts() -> int:
    """Parts per operand of the bf16 split images the current mode produces / consumes ("f16x3": the images GEMM and attention
    epilogues write are 3-part bf16; only the GroupNorm / LayerNorm producers write fp16 images there)."""
    return 2 if MMA_MODE == "bf16x3" else 3


def f16_mode() -> bool:
    return MMA_MODE == "f16x3"


F16_FF_OUT = os.environ.get("ALDM_F16_FF_OUT", "1") != "0"   # A/B switch: the GEGLU output as an fp16 image (FF-out in f16x3 too)
F16_ATTN = os.environ.get("ALDM_F16_ATTN", "1") != "0"       # A/B switch: self-attention on fp16 K / V^T images (three products)
F16_ATTN_OUT = os.environ.get("ALDM_F16_ATTN_OUT", "1") != "0"   # A/B switch: that attention's output as an fp16 image (to_out in f16x3)


def _pow2_scale(bound: float) -> float:
    """The largest power of two s with s * bound <= 32768 (a factor two under fp16's 65504: the bound is mathematical, the slack
    covers fp32 rounding of the normalisation itself)."""
    import math
    if not (bound > 0.0) or not math.isfinite(bound):
        return 1.0
    return float(2.0 ** math.floor(math.log2(32768.0 / bound)))


def _absmax_cached(t: Optional[torch.Tensor]) -> float:
    """max|t| of a parameter tensor, read once (one host sync, before any graph capture: every model runs an eager step first) and
    kept ON the tensor object together with its version counter (not keyed by data_ptr: the allocator reuses addresses)."""
    if t is None:
        return 0.0
    c = getattr(t, "_aldm_absmax", None)
    if c is None or c[0] != t._version:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("f16x3: the bound of a normalisation's parameters must be known before graph capture (run one eager step)")
        c = (t._version, float(t.detach().abs().max()))
        try:
            t._aldm_absmax = c
        except (AttributeError, RuntimeError):
            pass
    return c[1]


def _row_norm_bound(gamma: torch.Tensor, beta: Optional[torch.Tensor], C: int) -> float:
    """||gamma * x_hat + beta||_2 <= max|gamma| sqrt(C) + ||beta||_2 for a LayerNorm row (||x_hat||_2 <= sqrt(C))."""
    import math
    b2 = 0.0
    if beta is not None:
        c = getattr(beta, "_aldm_norm2", None)
        if c is None or c[0] != beta._version:
            c = (beta._version, float(beta.detach().double().norm()))
            try:
                beta._aldm_norm2 = c
            except (AttributeError, RuntimeError):
                pass
        b2 = c[1]
    return _absmax_cached(gamma) * math.sqrt(float(C)) + b2


def _norm_f16_scale(gamma: torch.Tensor, beta: Optional[torch.Tensor], n: int) -> float:
    """fp16 image scale of a GroupNorm / LayerNorm output normalised over n elements: |gamma x_hat + beta| <= sqrt(n) max|gamma| +
    max|beta| (|x_hat| <= sqrt(n) holds for any data; SiLU only shrinks)."""
    import math
    return _pow2_scale(math.sqrt(float(n)) * _absmax_cached(gamma) + _absmax_cached(beta))


@dataclass
class Packed:
    """A weight re-laid-out once for the igemm B operand: [ceil(K/4)][Npad][4], plus — in
    "bf16x6" mode — its bf16-split image [4*ceil(K/32)][3][Npad][8 bf16] (aldm_pack_split_bf16)."""
    data: torch.Tensor
    N: int
    Cin: int
    KH: int
    KW: int
    bias: Optional[torch.Tensor] = None
    split: Optional[torch.Tensor] = None
    split2: Optional[torch.Tensor] = None   # the 2-part ("bf16x3") image, DMA-fed launches only
    split16: Optional[torch.Tensor] = None  # the 2-part fp16 ("f16x3") image of w_scale * w
    w_scale: float = 0.0                    # ... its power-of-two scale
    cmax: Optional[float] = None            # largest column 2-norm (out_bound)
    bmax: float = 0.0

    @property
    def K(self) -> int:
        return self.KH * self.KW * self.Cin

    def out_bound(self, rn: float) -> float:
        """|x . w_n + b_n| <= rn * max_n ||w_n||_2 + max|b| for every row x with ||x||_2 <= rn (Cauchy-Schwarz); the weight's largest
        column norm and bias are read once (host sync, before graph capture) and cached."""
        if self.cmax is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("f16x3: a weight's column norms must be known before graph capture (run one eager step)")
            npad = self.data.numel() // (4 * ((self.K + 3) // 4))
            self.cmax = float(self.data.view(-1, npad, 4).double().pow(2).sum((0, 2)).max().sqrt())
            self.bmax = 0.0 if self.bias is None else float(self.bias.abs().max())
        return rn * self.cmax + self.bmax

    def split16_ptr(self) -> int:
        """Device pointer of the "f16x3" weight image (built on first use, before any graph capture): hi / lo fp16 of w_scale * w,
        w_scale = the largest power of two that keeps max|w| under 32768."""
        if self.split16 is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("a weight's fp16 split image must exist before graph capture (run one eager step)")
            lib = _l.load()
            self.w_scale = _pow2_scale(float(self.data.abs().max()))
            self.split16 = torch.empty(lib.aldm_split_bytes_parts(self.K, self.N, 2) // 4, device=self.data.device, dtype=torch.int32)
            _l.check(lib.aldm_pack_split_f16(self.data.data_ptr(), self.split16.data_ptr(), self.K, self.N, self.w_scale, _stream()),
                     "pack_split_f16")
        return self.split16.data_ptr()

    def split_ptr(self, parts: int = 3) -> Optional[int]:
        """Device pointer of the bf16-split image with `parts` parts (built on first use) or None in fp32 mode."""
        if MMA_MODE == "f32":
            return None
        if parts == 2:
            if self.split2 is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("a weight's 2-part split image must exist before graph capture (run one eager step)")
                lib = _l.load()
                self.split2 = torch.empty(lib.aldm_split_bytes_parts(self.K, self.N, 2) // 4, device=self.data.device,
                                          dtype=torch.int32)
                _l.check(lib.aldm_pack_split_bf16_parts(self.data.data_ptr(), self.split2.data_ptr(), self.K, self.N, 2,
                                                        _stream()), "pack_split_bf16(2)")
            return self.split2.data_ptr()
        if self.split is None:
            if torch.cuda.is_current_stream_capturing():
                return None  # never allocate inside a capture: this launch stays on the fp32 MFMA
            lib = _l.load()
            self.split = torch.empty(lib.aldm_split_bytes(self.K, self.N) // 4, device=self.data.device,
                                     dtype=torch.int32)
            _l.check(lib.aldm_pack_split_bf16(self.data.data_ptr(), self.split.data_ptr(), self.K, self.N,
                                              _stream()), "pack_split_bf16")
        return self.split.data_ptr()


class SplitT:
    """A split image (include/aldm_hip.h "split images"): the exact 3-way bf16 split of a channels-last fp32 tensor of
    logical shape `shape` = [..., C], stored [rows, C/32, 3, 32] as int16 — the pre-split A operand of the DMA-fed GEMM."""
    __slots__ = ("data", "shape", "fmt", "scale", "rn")

    def __init__(self, data: torch.Tensor, shape, fmt: str = "bf16", scale: float = 1.0, rn: float = 0.0):
        self.data = data
        self.shape = tuple(shape)
        self.fmt = fmt        # "bf16" | "f16" (the "f16x3" image: 2 parts, IEEE fp16, of scale * value)
        self.scale = scale
        self.rn = rn          # a-priori bound of a ROW's 2-norm (LayerNorm outputs: sqrt(C) max|gamma| + ||beta||), 0 = unknown:
                              # with it a projection's outputs are bounded by rn * (largest column norm of W) + max|bias|

    @property
    def parts(self) -> int:
        return self.data.shape[2]

    @staticmethod
    def empty(shape, device, parts: Optional[int] = None, f16_scale: float = 0.0) -> "SplitT":
        C_ = shape[-1]
        assert C_ % 32 == 0, f"split image needs C % 32 == 0, got {C_}"
        rows = 1
        for s_ in shape[:-1]:
            rows *= s_
        if f16_scale:
            return SplitT(torch.empty((rows, C_ // 32, 2, 32), device=device, dtype=torch.int16), shape, "f16", f16_scale)
        return SplitT(torch.empty((rows, C_ // 32, parts or split_parts(), 32), device=device, dtype=torch.int16), shape)

    @property
    def C(self) -> int:
        return self.shape[-1]

    @property
    def rows(self) -> int:
        return self.data.shape[0]

    @property
    def device(self):
        return self.data.device

    def data_ptr(self) -> int:
        return self.data.data_ptr()

    def view(self, *shape) -> "SplitT":
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else tuple(shape)
        n = 1
        for s_ in shape:
            n *= abs(s_)
        tot = self.rows * self.C
        shape = tuple(tot // n if s_ == -1 else s_ for s_ in shape)
        assert shape[-1] == self.C, "a split image can only be re-viewed over its row dimensions"
        return SplitT(self.data, shape, self.fmt, self.scale, self.rn)

    def float(self) -> torch.Tensor:
        """hi + mid + lo back to fp32 (tests / debugging): exact for bf16 images; (hi + lo) / scale for an fp16 image."""
        if self.fmt == "f16":
            h = self.data.view(torch.float16).float()
            return ((h[:, :, 0] + h[:, :, 1]) / self.scale).reshape(self.shape)
        parts = (self.data.to(torch.int32) << 16).view(torch.float32)
        acc = parts[:, :, 0] + parts[:, :, 1]
        if self.parts == 3:
            acc = acc + parts[:, :, 2]
        return acc.reshape(self.shape)


# DMA-fed GEMM path (pre-split activations, csrc/igemm_dma.h): on by default in "bf16x6" mode; ALDM_DMA=0 / set_dma(False)
# keep every activation fp32 and split A inside the K loop (the round-1 kernels).
DMA_MODE = os.environ.get("ALDM_DMA", "1") != "0"


def set_dma(on: bool) -> bool:
    global DMA_MODE
    prev, DMA_MODE = DMA_MODE, bool(on)
    return prev


def use_dma() -> bool:
    return DMA_MODE and MMA_MODE in ("bf16x6", "bf16x3", "f16x3")


def split_rows(x: torch.Tensor, x2: Optional[torch.Tensor] = None, pre=None, act: int = ACT_NONE,
               want_raw: bool = False, slope: float = 0.0):
    """split(act(x*scale + shift)) of channels-last x (++ x2 along C) -> SplitT; pre = (scale, shift) each [B, C]
    (GroupNorm apply, from gn_stats); act: ACT_NONE | ACT_SILU | ACT_LRELU(slope).  want_raw: also return split(x ++ x2)."""
    _chk(x, "split_rows.x")
    C1 = x.shape[-1]
    C2 = 0
    if x2 is not None:
        _chk(x2, "split_rows.x2")
        assert x2.shape[:-1] == x.shape[:-1]
        C2 = x2.shape[-1]
    shape = (*x.shape[:-1], C1 + C2)
    rows = x.numel() // C1
    P = rows // x.shape[0]
    dst = SplitT.empty(shape, x.device)
    raw = SplitT.empty(shape, x.device) if want_raw else None
    sc = sh = None
    if pre is not None:
        sc, sh = pre
        assert sc.shape == (x.shape[0], C1 + C2) and sc.is_contiguous() and sh.is_contiguous()
    _l.check(_l.load().aldm_split_rows_act(x.data_ptr(), _p(x2), C1, C2, rows, P, _p(sc), _p(sh), act, slope, dst.data_ptr(),
                                           None if raw is None else raw.data_ptr(), dst.parts, _stream()), "split_rows")
    return (dst, raw) if want_raw else dst


# Optional profiling hook (bench.py's roofline leg): when PROFILE is a list, every igemm launch is
# bracketed by events on the launch stream and recorded as (what, BM, BN, flops, ev_start, ev_end,
# (M, N, K, taps, C2, has_pre, pre_act, batch, splits)).
PROFILE = None
ATTN_PROFILE = None   # when a list: every ops.attention launch as (B, heads, Lq, Lk, masked, flops, ev_start, ev_end)


# Split-K scratch (caller-owned, see aldm_igemm_ws_floats): ONE grow-only buffer per device.  Launches
# are stream-ordered on the current stream, so consecutive igemm calls can share it.  It grows during
# the eager first DDIM step (every shape of the model is seen there), i.e. before graph capture.
_ws = {}
_ws_retired = []  # outgrown buffers stay alive: a captured HIP graph may still point at them
WS_SLOT = 0  # concurrent branches (pipeline.apply_model_cfg on two streams) use distinct scratch slots


def _workspace(lib, d: IgemmDesc, device) -> None:
    need = lib.aldm_igemm_ws_floats(C.byref(d))
    if need <= 0:
        return
    key = (device.index if device.index is not None else torch.cuda.current_device(), WS_SLOT)
    buf = _ws.get(key)
    if buf is None or buf.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            return  # never allocate inside a capture: the launch simply runs without split-K
        if buf is not None:
            _ws_retired.append(buf)
        buf = torch.empty(max(need, 8 << 20), device=device, dtype=torch.float32)
        _ws[key] = buf
    d.ws = buf.data_ptr()
    d.ws_floats = buf.numel()


# ---- tuned launch configurations -------------------------------------------------------------------
# audioldm2_amd/tuning/mi355x_igemm.json: {(geometry key): [BM, BN, splits]} measured on MI355X by
# tools/igemm_autotune.py for the shapes of the shipped model configs; everything else (and everything
# when ALDM_NO_TUNING=1) uses the library's built-in cost model.  Only speed and the (deterministic)
# fp32 summation order depend on the choice.
_TUNE_FIELDS = ("B", "H", "W", "C1", "C2", "pix1", "up_h", "up_w", "KH", "KW", "SH", "SW", "PH", "PW", "DH", "DW",
                "OH", "OW", "N", "b_mode", "batch", "epi_mode", "out_mul")
_TUNED = None
TUNE_LOG = None  # when a list: every launch's geometry key is appended (tools/igemm_autotune.py)


def tune_key(d: IgemmDesc) -> str:
    """Geometry key of a launch (tuned tables, TUNE_LOG); DMA-fed launches (pre-split A operand) carry a ",dma" suffix:
    they have their own kernel family and table."""
    return ",".join(str(getattr(d, f)) for f in _TUNE_FIELDS) + f",{_pre_mode(d)}" + \
        (("," + ("dma2" if d.split_parts == 2 else "dma")) if d.a_split else "")


def _tuned_table(bx=False):
    """{geometry key: [BM, BN, splits, kgroups(, mma)]} for the fp32-MFMA (bx=False) or bf16-split (True) launches;
    bx="dma": {key: [BM, BN, splits, stages]} for the DMA-fed kernel."""
    global _TUNED
    if _TUNED is None:
        _TUNED = {False: {}, True: {}, "dma": {}, "dma2": {}}
        if os.environ.get("ALDM_NO_TUNING", "0") != "1":
            for mode, name in ((False, "mi355x_igemm.json"), (True, "mi355x_igemm_bf16x6.json"),
                               ("dma", "mi355x_igemm_dma.json"), ("dma2", "mi355x_igemm_dma_bf16x3.json")):
                # ($ALDM_TUNING_DIR: another directory of tables with the same file names — same-box A/Bs of two tunings;
                #  a table missing there falls back to the shipped one)
                path = os.path.join(os.environ.get("ALDM_TUNING_DIR", ""), name)
                if not (os.environ.get("ALDM_TUNING_DIR") and os.path.exists(path)):
                    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", name)
                if os.path.exists(path):
                    with open(path) as f:
                        _TUNED[mode] = {k: v[:5] if mode is True else v[:4] for k, v in json.load(f)["entries"].items()}
    return _TUNED[bx]


def _set_split_operand(d: IgemmDesc, x: "SplitT", pw: "Packed", so: Optional["SplitT"] = None) -> None:
    """a_split / w_split / image formats of a DMA-fed launch over the split image x (and its split-image output so)."""
    d.a_split = x.data_ptr()
    d.split_parts = x.parts
    if x.fmt == "f16":
        d.a_fmt = _l.FMT_F16
        d.w_split = pw.split16_ptr()
        d.acc_scale = 1.0 / (x.scale * pw.w_scale)
        d.out_split_parts = so.parts if so is not None else 3
    else:
        d.w_split = pw.split_ptr(x.parts)
        if so is not None:
            assert so.parts == x.parts and so.fmt == "bf16"


def _igemm(d: IgemmDesc, what: str, device=None):
    lib = _l.load()
    if d.a_split and d.epi_mode == _l.EPI_QKV:   # tuned (and logged) like the plain projection of the same geometry
        d.epi_mode = _l.EPI_PLAIN
        key = tune_key(d)
        d.epi_mode = _l.EPI_QKV
    else:
        key = tune_key(d)
    key_t = key
    if TUNE_LOG is not None:
        TUNE_LOG.append(key)
    if d.a_split:
        hint = _tuned_table("dma2" if d.split_parts == 2 else "dma").get(key_t)
        if hint is not None:
            d.hint_bm, d.hint_bn, d.hint_splits, d.hint_stages = hint[:4]
    else:
        hint = _tuned_table(bool(d.w_split)).get(key)
        if hint is not None:
            d.hint_bm, d.hint_bn, d.hint_splits, d.hint_kgroups = hint[:4]
            d.hint_mma = hint[4] if len(hint) > 4 else 0  # bf16x6 table: 1 = this shape is faster on the fp32 MFMA
    _workspace(lib, d, device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    if PROFILE is None:
        _l.check(lib.aldm_igemm(C.byref(d), _stream()), what)
        return
    bm, bn, fl, sp, kg, mma = C.c_int(), C.c_int(), C.c_int64(), C.c_int(), C.c_int(), C.c_int()
    _l.check(lib.aldm_igemm_plan(C.byref(d), C.byref(bm), C.byref(bn), C.byref(fl), C.byref(sp), C.byref(kg),
                                 C.byref(mma)), what)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    _l.check(lib.aldm_igemm(C.byref(d), _stream()), what)
    e1.record()
    shape = (d.B * d.OH * d.OW, d.N, d.K, d.KH * d.KW, d.C2, int(bool(d.pre_scale)), d.pre_act, d.batch,
             sp.value * 10 + kg.value, int(bool(d.a_split)), int(bool(d.out)), int(bool(d.out_split)), d.split_parts or 3,
             int(d.epi_mode == _l.EPI_GEGLU), int(bool(d.res)))
    PROFILE.append((what, bm.value, bn.value, fl.value, e0, e1, shape,
                    _kernel_name(d, bm.value, bn.value, kg.value, bool(mma.value))))


def _pre_mode(d: IgemmDesc) -> int:
    """Prologue mode a descriptor dispatches to (PRE_* of csrc/igemm_kernel.h)."""
    if not d.pre_scale and d.pre_act == ACT_NONE:
        return 0
    if d.pre_scale and d.pre_act == ACT_NONE:
        return 1
    if d.pre_scale and d.pre_act == ACT_SILU:
        return 2
    if not d.pre_scale and d.pre_act == ACT_LRELU:
        return 3
    return 4


def _kernel_name(d: IgemmDesc, bm: int, bn: int, kg: int = 1, bx: bool = False) -> str:
    """The igemm instantiation a descriptor dispatches to, spelled like rocprofv3's kernel names
    (igemm_kernel<BM, BN, WM, WN, PRE, KGRP, UNI, BX>)."""
    if d.a_split:
        nst = _l.load().aldm_igemm_plan_stages(C.byref(d))
        if nst >= 400:   # halo-patch 3x3 convolution (csrc/igemm_dma_halo.h): 400 + 10 * (8 waves) + weight-ring depth; rocprofv3
            # appends the patch capacity and the fp16 flag: <BM, BN, ring, WM, parts, MAXCH, F16>
            w8 = (nst - 400) // 10
            return f"igemm_dma_halo_kernel<{bm}, {bn}, {(nst - 400) % 10}, {4 if (bm == 256 or w8) else 2}, {d.split_parts or 3}>"
        if nst >= 300:   # operand-stationary kernel for short K (csrc/igemm_dma_os.h): <k-tiles, ring depth, parts, epilogue form>
            epi = 1 if d.epi_mode == _l.EPI_GEGLU else (2 if d.epi_mode == _l.EPI_QKV else 0)   # OS_EPI_GEGLU / _QKV / _PLAIN
            return f"igemm_dma_os_kernel<{d.K // 32}, {nst - 300}, {d.split_parts or 3}, {epi}>"
        if nst >= 200:   # loader waves (csrc/igemm_dma_lw.h; rocprofv3 appends the blocks-per-CU template argument)
            return f"igemm_dma_lw_kernel<{bm}, {bn}, {nst - 200}, {4 if bm == 256 else 2}, {d.split_parts or 3}>"
        if nst >= 100:   # the persistent wave-specialised form (csrc/igemm_dma_ws.h)
            return f"igemm_dma_ws_kernel<{bm}, {bn}, {nst - 100}, {d.split_parts or 3}>"
        return f"igemm_dma_kernel<{bm}, {bn}, {nst}, {4 if bm == 256 else 2}, {d.split_parts or 3}>"
    pre = _pre_mode(d)
    wm, wn = (4, 1) if bn == 32 else (2, 2)
    # 8 waves per tile: same rule as csrc/igemm.hip (ALDM_IGEMM_W8 tile mask, default 128x128 GroupNorm prologues)
    w8 = _wave8_mask if _wave8_mask >= 0 else int(os.environ.get("ALDM_IGEMM_W8", "1"))
    bit = {(128, 128): 1, (64, 128): 2, (128, 64): 4}.get((bm, bn), 0)
    if kg == 1 and d.epi_mode != _l.EPI_GEGLU and (w8 & bit) and (
            bit == 1 if bx else (pre in (1, 2) or (bit == 1 and (w8 & 8)))):
        wm, wn = (4, 2) if bit == 4 else (2, 4)
    if bx and (bm, bn) == (128, 128) and d.epi_mode == _l.EPI_GEGLU and pre == 0:
        wm, wn = 4, 2  # 8 waves as 4x2 for the GEGLU epilogue (csrc/igemm_bx_pre0.hip)
    uni = "true" if pre in (1, 2) and (d.OH * d.OW) % bm == 0 else "false"
    return f"igemm_kernel<{bm}, {bn}, {wm}, {wn}, {pre}, {kg}, {uni}, {'true' if bx else 'false'}>"


def igemm_force(bm: int = 0, bn: int = 0, splits: int = 0, kgroups: int = 0, stages: int = 0) -> None:
    """Tuning override for tools/tests: force tile / split-K / wave groups (/ LDS ring depth of the DMA-fed kernel) of
    subsequent igemm launches (bm = 0: automatic)."""
    _l.load().aldm_igemm_force(bm, bn, splits, kgroups)
    _l.load().aldm_igemm_force_stages(stages)


def attention_mma(mode: int) -> int:
    """Matrix-core path of ops.attention (aldm_attention_mma): 1 fp32 MFMA, 2 bf16-split, -1 default.  Returns the
    previous mode."""
    return _l.load().aldm_attention_mma(mode)


def attention_sched(sched: int) -> int:
    """Schedule of the pre-split self-attention kernel (aldm_attention_sched): -1 default, 0 round-3/4 pipelined, 1 re-scheduled
    exact-max loop, 2 one-pass fixed-reference loop, 3 K / V^T through LDS once per block.  Returns the previous setting."""
    return _l.load().aldm_attention_sched(sched)


def debug_drop_product(on: bool) -> bool:
    """TEST HOOK (aldm_debug_drop_product): DMA-fed launches leave out the smallest of the six bf16 partial products — only the
    classic 64x128 / 2-stage tile has that instantiation, every other igemm launch fails while the switch is on.  Returns the
    previous setting.  Used by tests/test_dma_gpu.py to show the fp32-grade bar catches a lost product.  The symbol exists only in the
    -DALDM_TEST_HOOKS variant library (libaldm_hip_testhooks.so, selected with $ALDM_LIB_PATH): on the release library this raises."""
    fn = getattr(_l.load(), "aldm_debug_drop_product", None)
    if fn is None:
        raise RuntimeError("aldm_debug_drop_product is a test hook of libaldm_hip_testhooks.so; the release library does not "
                           f"export it (run with ALDM_LIB_PATH={_l.TESTHOOKS_LIB_PATH})")
    return bool(fn(1 if on else 0))


def igemm_mma(mode: int) -> int:
    """Tuning override for tools/tests (aldm_igemm_mma): 0 automatic, 1 fp32 MFMA always, 2 bf16-split wherever an
    instantiation exists.  Returns the previous mode."""
    return _l.load().aldm_igemm_mma(mode)


_wave8_mask = -1


def igemm_wave8(mask: int = -1) -> int:
    """Tuning override for tools/tests: which block tiles run with 8 wavefronts per tile (bit mask, see
    aldm_igemm_wave8_mask in include/aldm_hip.h); mask < 0 restores the default."""
    global _wave8_mask
    _wave8_mask = mask
    return _l.load().aldm_igemm_wave8_mask(mask)


def _npad(n: int) -> int:
    return (n + 31) // 32 * 32


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> Packed:
    """weight: Linear [N, Cin], Conv1d [N, Cin, KW] or Conv2d [N, Cin, KH, KW] (PyTorch layouts)."""
    w = weight.detach().to(device="cuda", dtype=torch.float32).contiguous()
    if w.dim() == 2:
        N, Cin, KH, KW = w.shape[0], w.shape[1], 1, 1
    elif w.dim() == 3:
        N, Cin, KH, KW = w.shape[0], w.shape[1], 1, w.shape[2]
    else:
        N, Cin, KH, KW = w.shape
    K = KH * KW * Cin
    dst = torch.empty(((K + 3) // 4) * _npad(N) * 4, device="cuda", dtype=torch.float32)
    lib = _l.load()
    _l.check(lib.aldm_pack_weight(w.data_ptr(), dst.data_ptr(), N, Cin, KH, KW, 0, 0, 1, _stream()),
             "pack_weight")
    b = None if bias is None else bias.detach().to(device="cuda", dtype=torch.float32).contiguous()
    return Packed(dst, N, Cin, KH, KW, b)


def pack_geglu(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> Packed:
    """GEGLU projection Linear(dim, 2*inner) (attention.py:40): rows [0, inner) are the value half, rows
    [inner, 2*inner) the gate half (`x, gate = proj(x).chunk(2, -1)`).  Re-order the output channels into
    groups of 32 value rows followed by their 32 gate rows so one MFMA wave holds both (EPI_GEGLU)."""
    two_inner = weight.shape[0]
    inner = two_inner // 2
    assert weight.dim() == 2 and two_inner % 64 == 0
    g = torch.arange(inner // 32).view(-1, 1, 1)
    j = torch.arange(2).view(1, -1, 1)
    l = torch.arange(32).view(1, 1, -1)
    perm = (j * inner + g * 32 + l).reshape(-1).to(weight.device)
    return pack_conv(weight.detach()[perm], None if bias is None else bias.detach()[perm])


def linear_geglu(x, pw: Packed, split_out: Optional[str] = None, gate_act: int = ACT_NONE):
    """y = value * gelu_erf(gate) with [value | gate] = x @ W^T + b fused into the GEMM epilogue
    (attention.py:37-45); pw from pack_geglu.  x: [..., Cin] fp32 or SplitT -> [..., N/2] (split_out: None -> fp32,
    "only" -> SplitT, "also" -> (fp32, SplitT))."""
    is_split = isinstance(x, SplitT)
    if not is_split:
        _chk(x, "linear_geglu.x")
    shp = x.shape
    M = (x.rows if is_split else x.numel() // shp[-1])
    assert shp[-1] == pw.Cin and pw.KH == 1 and pw.KW == 1 and pw.N % 64 == 0
    oshape = (*shp[:-1], pw.N // 2)
    out = None if split_out == "only" else torch.empty(oshape, device=x.device, dtype=torch.float32)
    so = None
    if split_out:
        if is_split and x.fmt == "f16" and x.rn > 0.0 and f16_mode() and F16_FF_OUT:
            # "f16x3" all the way through the MLP: |value * gelu(gate)| <= |value| |gate| <= (rn c + b)^2 — the GEGLU output has an
            # a-priori bound too, so it is written as an fp16 image and the FF-out GEMM runs three products as well
            so = SplitT.empty(oshape, x.device, f16_scale=_pow2_scale(pw.out_bound(x.rn) ** 2))
        else:
            so = SplitT.empty(oshape, x.device)
    d = IgemmDesc()
    if is_split:
        _set_split_operand(d, x, pw, so if (so is None or so.fmt == "bf16") else None)
        if so is not None and so.fmt == "f16":
            d.out_split_fmt = _l.FMT_F16; d.out_split_scale = so.scale
    else:
        d.x1 = x.data_ptr(); d.split_parts = 3
        d.w_split = pw.split_ptr(3)
    if so is not None and not is_split:
        d.split_parts = so.parts
    d.C1 = pw.Cin; d.B = 1; d.H = 1; d.W = M; d.up_h = d.up_w = 1
    d.KH = d.KW = d.SH = d.SW = d.DH = d.DW = 1
    d.OH = 1; d.OW = M
    d.w = pw.data.data_ptr(); d.b_mode = B_PACKED; d.K = pw.K; d.N = pw.N
    d.bias = _p(pw.bias); d.out = _p(out); d.ldo = pw.N // 2; d.alpha = 1.0
    if so is not None:
        d.out_split = so.data_ptr(); d.out_split_c = pw.N // 2
    d.epi_mode = _l.EPI_GEGLU; d.batch = 1
    d.act = gate_act  # ACT_GELU_TANH: tanh-GELU gate (T5 gated-gelu FF); default erf GELU (attention.py:44)
    _igemm(d, "igemm(geglu)")
    return so if split_out == "only" else ((out, so) if split_out else out)


def linear_qkv(x: "SplitT", pw: Packed, heads: int, rows_per_sample: int):
    """The fused self-attention projection [q | k | v] = x W^T (pw: the three weights concatenated along N, no bias) with the
    ALDM_EPI_QKV epilogue: returns (q fp32 [..., C], k_img, vt_img) — k as the split image of its columns, v transposed per
    (sample, head, 32-key tile) straight from the accumulators — the operands ops.attention_presplit multiplies without
    splitting anything in its key loop.  x: a SplitT [B, L, C_in] (DMA-fed launch)."""
    assert isinstance(x, SplitT) and pw.KH == 1 and pw.KW == 1 and pw.bias is None
    Cq = heads * 32
    assert pw.N == 3 * Cq and x.shape[-1] == pw.Cin
    M = x.rows
    assert M % rows_per_sample == 0 and rows_per_sample % 32 == 0
    Bn = M // rows_per_sample
    # "f16x3": q, k, v of a LayerNorm-fed projection are bounded by R c (R the rows' 2-norm bound, c the weight's largest column
    # norm) — K and V^T are written as fp16 images under that bound and the attention runs three products too; without a bound
    # (or with the switch off) they are 3-part bf16 images and the attention runs bf16x6 behind the f16x3 projection
    f16_kv = x.fmt == "f16" and x.rn > 0.0 and f16_mode() and F16_ATTN
    P = 2 if f16_kv else (3 if x.fmt == "f16" else x.parts)
    dev = x.device
    q = torch.empty((*x.shape[:-1], Cq), device=dev, dtype=torch.float32)
    k_img = torch.empty((M, heads, P, 32), device=dev, dtype=torch.int16)
    vt_img = torch.empty((Bn, heads, rows_per_sample // 32, P, 32, 32), device=dev, dtype=torch.int16)
    d = IgemmDesc()
    _set_split_operand(d, x, pw)
    d.out_split_parts = P
    if f16_kv:
        bound = pw.out_bound(x.rn)
        kv_scale = _pow2_scale(bound)
        q_scale = _pow2_scale(bound * (32 ** -0.5) * 1.4426950408889634)
        d.out_split_fmt = _l.FMT_F16; d.out_split_scale = kv_scale; d.vt_scale = kv_scale
        k_img._aldm_f16 = (q_scale, kv_scale, kv_scale)
    d.C1 = pw.Cin; d.B = 1; d.H = 1; d.W = M; d.up_h = d.up_w = 1
    d.KH = d.KW = d.SH = d.SW = d.DH = d.DW = 1
    d.OH = 1; d.OW = M
    d.w = pw.data.data_ptr(); d.b_mode = B_PACKED; d.K = pw.K; d.N = pw.N
    d.out = q.data_ptr(); d.ldo = Cq; d.alpha = 1.0
    d.k_split = k_img.data_ptr(); d.vt_split = vt_img.data_ptr(); d.qkv_c = Cq; d.qkv_rows = rows_per_sample
    d.epi_mode = _l.EPI_QKV; d.batch = 1
    _igemm(d, "igemm(qkv)")
    return q, k_img, vt_img


def attention_presplit(q: torch.Tensor, k_img: torch.Tensor, vt_img: torch.Tensor, heads: int, *,
                       scale: Optional[float] = None, split_out: Optional[str] = None):
    """Self-attention over the operands linear_qkv wrote (aldm_attention_d32_presplit): bit-identical to
    ops.attention(q, k, v) in the same mode, without the per-key-tile operand splits."""
    qp, Lq, ldq = _rowview(q, "attn_pre.q")
    B = q.shape[0]
    Lk = k_img.shape[0] // B
    parts = k_img.shape[2]
    assert q.shape[2] == heads * 32 and k_img.shape[1] == heads and vt_img.shape[0] == B and vt_img.shape[2] * 32 == Lk
    if scale is None:
        scale = 32 ** -0.5
    out = None if split_out == "only" else torch.empty((B, Lq, heads * 32), device=q.device, dtype=torch.float32)
    f16s = getattr(k_img, "_aldm_f16", None)
    # fp16 K / V^T images: the output is a convex combination of the values (|out| <= max|v|), so its image can be an fp16 one
    # under V's scale and the to_out projection runs three products too (ALDM_F16_ATTN_OUT=0: a 3-part bf16 image, six products)
    f16_out = f16s is not None and F16_ATTN_OUT
    so = None
    if split_out:
        so = SplitT.empty((B, Lq, heads * 32), q.device, f16_scale=f16s[2]) if f16_out else \
            SplitT.empty((B, Lq, heads * 32), q.device, 3 if f16s is not None else parts)
    ev = None
    if ATTN_PROFILE is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if f16s is not None:   # fp16 K / V^T images (linear_qkv in "f16x3" mode): three-product attention, q split under q_scale
        assert abs(scale - 32 ** -0.5) < 1e-12, "the fp16 images' q scale was chosen for the default softmax scale"
        _l.check(_l.load().aldm_attention_d32_presplit_f16(qp, k_img.data_ptr(), vt_img.data_ptr(), _p(out),
                                                           None if so is None else so.data_ptr(),
                                                           3 if so is None else (0 if so.fmt == "f16" else so.parts),
                                                           0.0 if so is None or so.fmt != "f16" else so.scale, B, heads,
                                                           Lq, Lk, ldq, heads * 32, scale, f16s[0], f16s[1], f16s[2], _stream()),
                 "attention_d32_presplit_f16")
    else:
        _l.check(_l.load().aldm_attention_d32_presplit(qp, k_img.data_ptr(), vt_img.data_ptr(), _p(out),
                                                       None if so is None else so.data_ptr(), parts, B, heads, Lq, Lk, ldq,
                                                       heads * 32, scale, _stream()), "attention_d32_presplit")
    if ev is not None:
        ev[1].record()
        ATTN_PROFILE.append((B, heads, Lq, Lk, False, 4.0 * B * heads * Lq * Lk * 32, ev[0], ev[1]))
    if not split_out:
        return out
    return so if split_out == "only" else (out, so)


def pack_convtr1d(weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int) -> List[Packed]:
    """ConvTranspose1d weight [Cin, N, K] -> one Packed per output phase (polyphase decomposition)."""
    w = weight.detach().to(device="cuda", dtype=torch.float32).contiguous()
    Cin, N, KW = w.shape
    T = (KW + stride - 1) // stride
    lib = _l.load()
    b = None if bias is None else bias.detach().to(device="cuda", dtype=torch.float32).contiguous()
    out = []
    for ph in range(stride):
        dst = torch.empty(((T * Cin + 3) // 4) * _npad(N) * 4, device="cuda", dtype=torch.float32)
        _l.check(lib.aldm_pack_weight(w.data_ptr(), dst.data_ptr(), N, Cin, 1, KW, 1, ph, stride,
                                      _stream()), "pack_weight(tr)")
        out.append(Packed(dst, N, Cin, 1, T, b))
    return out


def conv(x: torch.Tensor, pw: Packed, *, stride=(1, 1), pad=(0, 0), dil=(1, 1), up=(1, 1),
         x2: Optional[torch.Tensor] = None, out_hw: Optional[Tuple[int, int]] = None,
         pre: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, pre_act: int = ACT_NONE,
         pre_slope: float = 0.0, bias: Optional[torch.Tensor] = None,
         rowbias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
         act: int = ACT_NONE, act_slope: float = 0.0, alpha: float = 1.0,
         out: Optional[torch.Tensor] = None, accumulate: bool = False,
         remap: Optional[Tuple[int, int, int]] = None,
         use_pw_bias: bool = True, split_out: Optional[str] = None, split_act: int = ACT_NONE, split_slope: float = 0.0):
    """Implicit-GEMM convolution (aldm_igemm).  x: [B, H, W, C1] (+ x2: [B, H, W, C2] concatenated
    along C), or a SplitT of that shape (pre-split operand -> DMA-fed kernel; no x2 / pre then).  Returns
    [B, OH, OW, N] (or the remapped [B, 1, out_len, N]); split_out = "only": a SplitT of the result instead,
    "also": (fp32, SplitT)."""
    is_split = isinstance(x, SplitT)
    if is_split:
        assert x2 is None and pre is None and pre_act == ACT_NONE, "a pre-split operand takes no prologue"
    else:
        _chk(x, "conv.x")
    B, H, W, C1 = x.shape
    C2 = 0
    if x2 is not None:
        _chk(x2, "conv.x2")
        assert x2.shape[:3] == x.shape[:3]
        C2 = x2.shape[3]
    assert C1 + C2 == pw.Cin, f"channels {C1}+{C2} != packed Cin {pw.Cin}"
    VH, VW = H * up[0], W * up[1]
    if out_hw is None:
        OH = (VH + 2 * pad[0] - dil[0] * (pw.KH - 1) - 1) // stride[0] + 1
        OW = (VW + 2 * pad[1] - dil[1] * (pw.KW - 1) - 1) // stride[1] + 1
    else:
        OH, OW = out_hw
    N = pw.N
    if remap is not None:
        out_mul, out_off, out_len = remap
        oshape = (B, 1, out_len, N)
    else:
        out_mul = out_off = out_len = 0
        oshape = (B, OH, OW, N)
    if split_out == "only":
        assert out is None and not accumulate
    elif out is None:
        assert not accumulate
        out = torch.empty(oshape, device=x.device, dtype=torch.float32)
    else:
        _chk(out, "conv.out")
        assert out.numel() == oshape[0] * oshape[1] * oshape[2] * oshape[3], (out.shape, oshape)
    so = SplitT.empty(oshape, x.device) if split_out else None
    if bias is None and use_pw_bias:
        bias = pw.bias
    d = IgemmDesc()
    if is_split:
        _set_split_operand(d, x, pw, so)
    else:
        d.x1 = x.data_ptr(); d.split_parts = so.parts if so is not None else 3
    d.x2 = _p(x2)
    d.C1 = C1; d.C2 = C2; d.pix1 = 0; d.pix2 = 0
    d.B = B; d.H = H; d.W = W; d.up_h = up[0]; d.up_w = up[1]
    d.KH = pw.KH; d.KW = pw.KW; d.SH = stride[0]; d.SW = stride[1]
    d.PH = pad[0]; d.PW = pad[1]; d.DH = dil[0]; d.DW = dil[1]
    d.OH = OH; d.OW = OW
    if pre is not None:
        d.pre_scale = pre[0].data_ptr(); d.pre_shift = pre[1].data_ptr()
    d.pre_act = pre_act; d.pre_slope = pre_slope
    d.w = pw.data.data_ptr(); d.b_mode = B_PACKED; d.ldb = 0
    # register-staged launches always use the 3-part weight image; a 2-part out_split from them needs split_parts = 2,
    # which rules the bf16-split register-staged kernel out (fp32 MFMA then): only DMA-fed launches write 2-part images
    if not is_split:
        d.w_split = pw.split_ptr(3)
    d.K = pw.K; d.N = N
    d.bias = _p(bias); d.rowbias = _p(rowbias); d.res = _p(res); d.out = _p(out)
    if so is not None:
        d.out_split = so.data_ptr(); d.out_split_c = N
        d.out_split_act = split_act; d.out_split_slope = split_slope   # activation on the split-image output only
    if rowbias is not None:
        if not (rowbias.dim() == 2 and rowbias.stride(1) == 1 and rowbias.shape == (B, N)):
            raise RuntimeError("conv.rowbias: need a [B, N] fp32 view with unit inner stride")
        d.rowbias_ld = rowbias.stride(0)
    d.ldo = N; d.act = act; d.act_slope = act_slope; d.alpha = alpha
    d.accumulate = 1 if accumulate else 0
    d.out_mul = out_mul; d.out_off = out_off; d.out_len = out_len
    d.batch = 1
    _igemm(d, "igemm(conv)")
    if split_out == "only":
        return so
    return (out.view(oshape), so) if split_out else out.view(oshape)


def linear(x, pw: Packed, **kw):
    """x: [..., Cin] (fp32 tensor or SplitT) -> [..., N] through the same engine (1x1 'conv' over M rows)."""
    shp = x.shape
    if isinstance(x, SplitT):
        M = x.rows
        y = conv(x.view(1, 1, M, shp[-1]), pw, **kw)
    else:
        M = x.numel() // shp[-1]
        y = conv(x.reshape(1, 1, M, shp[-1]), pw, **kw)
    oshape = (*shp[:-1], pw.N)
    if isinstance(y, tuple):
        return y[0].view(oshape), y[1].view(oshape)
    return y.view(oshape)


def _view3(t: torch.Tensor, name: str):
    """[Z, R, D] fp32 CUDA view with unit inner stride (a column slice of a wider buffer is fine)."""
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.stride(2) == 1):
        raise RuntimeError(f"{name}: need a [Z, R, D] fp32 CUDA view with unit inner stride")
    return t.data_ptr(), t.stride(0), t.stride(1)


def gemm_nt(a: torch.Tensor, bmat: torch.Tensor, *, alpha: float = 1.0,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Batched C[z] = alpha * A[z] @ Bmat[z]^T with A [Z, M, K], Bmat [Z, N, K] views (activation x
    activation product: Q K^T of the VAE mid attention, model.py:219)."""
    ap, sa, lda = _view3(a, "gemm_nt.a")
    bp, sb, ldb = _view3(bmat, "gemm_nt.b")
    Z, M, K = a.shape
    Zb, N, Kb = bmat.shape
    assert Z == Zb and K == Kb and K % 4 == 0
    if out is None:
        out = torch.empty((Z, M, N), device=a.device, dtype=torch.float32)
    d = IgemmDesc()
    d.x1 = ap; d.C1 = K; d.pix1 = lda; d.B = 1; d.H = 1; d.W = M; d.up_h = d.up_w = 1
    d.KH = d.KW = d.SH = d.SW = d.DH = d.DW = 1
    d.OH = 1; d.OW = M
    d.w = bp; d.b_mode = B_NT; d.ldb = ldb; d.K = K; d.N = N
    d.out = out.data_ptr(); d.ldo = N; d.alpha = alpha
    d.batch = Z; d.stride_x = sa; d.stride_w = sb; d.stride_o = M * N
    _igemm(d, "igemm(nt)")
    return out


def pack_kn(src: torch.Tensor) -> torch.Tensor:
    """[Z, K, N] activations (row pitch may exceed N) -> packed B operand [Z][ceil(K/4)][Npad][4]."""
    sp, ss, lds = _view3(src, "pack_kn.src")
    Z, K, N = src.shape
    per = ((K + 3) // 4) * _npad(N) * 4
    dst = torch.empty((Z, per), device=src.device, dtype=torch.float32)
    _l.check(_l.load().aldm_pack_kn(sp, dst.data_ptr(), K, N, lds, Z, ss, per, _stream()), "pack_kn")
    return dst


def gemm_packed_batched(a: torch.Tensor, bp: torch.Tensor, K: int, N: int, *,
                        out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Batched C[z] = A[z] @ B[z], B pre-packed by pack_kn (P V of the VAE mid attention)."""
    _chk(a, "gemm_packed.a")
    Z, M, Ka = a.shape
    assert Ka == K and K % 4 == 0
    if out is None:
        out = torch.empty((Z, M, N), device=a.device, dtype=torch.float32)
    d = IgemmDesc()
    d.x1 = a.data_ptr(); d.C1 = K; d.B = 1; d.H = 1; d.W = M; d.up_h = d.up_w = 1
    d.KH = d.KW = d.SH = d.SW = d.DH = d.DW = 1
    d.OH = 1; d.OW = M
    d.w = bp.data_ptr(); d.b_mode = B_PACKED; d.ldb = 0; d.K = K; d.N = N
    d.out = out.data_ptr(); d.ldo = N; d.alpha = 1.0
    d.batch = Z; d.stride_x = M * K; d.stride_w = bp.shape[1]; d.stride_o = M * N
    _igemm(d, "igemm(packed batched)")
    return out


_gn_ws = {}


def gn_stats(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, groups: int = 32,
             eps: float = 1e-5, x2: Optional[torch.Tensor] = None):
    """GroupNorm statistics of channels-last x (++ x2) -> (scale, shift), each [B, C]."""
    _chk(x, "gn.x")
    B = x.shape[0]
    C1 = x.shape[-1]
    P = x.numel() // (B * C1)
    C2 = 0
    if x2 is not None:
        _chk(x2, "gn.x2")
        C2 = x2.shape[-1]
    Cc = C1 + C2
    lib = _l.load()
    nws = lib.aldm_gn_ws_floats(B, P, Cc, groups)
    ws = torch.empty(nws, device=x.device, dtype=torch.float32)
    ss = torch.empty((2, B, Cc), device=x.device, dtype=torch.float32)
    _l.check(lib.aldm_groupnorm_stats(x.data_ptr(), _p(x2), B, P, C1, C2, groups, eps,
                                      gamma.data_ptr(), beta.data_ptr(), ss[0].data_ptr(),
                                      ss[1].data_ptr(), ws.data_ptr(), _stream()), "groupnorm_stats")
    return ss[0], ss[1]


def gn_split(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, groups: int = 32, eps: float = 1e-5,
             x2: Optional[torch.Tensor] = None, act: int = ACT_NONE, want_raw: bool = False):
    """split(act(GroupNorm(x ++ x2))) as a SplitT — gn_stats + split_rows in one call (one launch up to 1024 pixels per
    sample, aldm_groupnorm_split); want_raw: also split(x ++ x2)."""
    _chk(x, "gn_split.x")
    B = x.shape[0]
    C1 = x.shape[-1]
    P = x.numel() // (B * C1)
    C2 = 0
    if x2 is not None:
        _chk(x2, "gn_split.x2")
        assert x2.shape[:-1] == x.shape[:-1]
        C2 = x2.shape[-1]
    Cc = C1 + C2
    lib = _l.load()
    ws = torch.empty(lib.aldm_gn_ws_floats(B, P, Cc, groups), device=x.device, dtype=torch.float32)
    ss = torch.empty((2, B, Cc), device=x.device, dtype=torch.float32)
    shape = (*x.shape[:-1], Cc)
    raw = SplitT.empty(shape, x.device) if want_raw else None
    if f16_mode():
        # "f16x3": the normalised (and SiLU'd) tensor as a 2-part fp16 image; its bound is known before the data
        dst = SplitT.empty(shape, x.device, f16_scale=_norm_f16_scale(gamma, beta, (Cc // groups) * P))
        _l.check(lib.aldm_groupnorm_split_f16(x.data_ptr(), _p(x2), B, P, C1, C2, groups, eps, gamma.data_ptr(), beta.data_ptr(), act,
                                              ss[0].data_ptr(), ss[1].data_ptr(), ws.data_ptr(), dst.data_ptr(),
                                              None if raw is None else raw.data_ptr(), 3, dst.scale, _stream()), "groupnorm_split_f16")
        return (dst, raw) if want_raw else dst
    dst = SplitT.empty(shape, x.device)
    _l.check(lib.aldm_groupnorm_split(x.data_ptr(), _p(x2), B, P, C1, C2, groups, eps, gamma.data_ptr(), beta.data_ptr(), act,
                                      ss[0].data_ptr(), ss[1].data_ptr(), ws.data_ptr(), dst.data_ptr(),
                                      None if raw is None else raw.data_ptr(), dst.parts, _stream()), "groupnorm_split")
    return (dst, raw) if want_raw else dst


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              split_out: Optional[str] = None):
    """LayerNorm over the last dim; split_out = "only": the result as a SplitT (the next GEMM's pre-split operand),
    "also": (fp32, SplitT)."""
    _chk(x, "ln.x")
    Cc = x.shape[-1]
    M = x.numel() // Cc
    y = None if split_out == "only" else torch.empty_like(x)
    if not split_out:
        _l.check(_l.load().aldm_layernorm(x.data_ptr(), y.data_ptr(), M, Cc, gamma.data_ptr(),
                                          beta.data_ptr(), eps, _stream()), "layernorm")
        return y
    if f16_mode():
        so = SplitT.empty(x.shape, x.device, f16_scale=_norm_f16_scale(gamma, beta, Cc))
        so.rn = _row_norm_bound(gamma, beta, Cc)
        _l.check(_l.load().aldm_layernorm_split_f16(x.data_ptr(), _p(y), so.data_ptr(), M, Cc, gamma.data_ptr(), beta.data_ptr(), eps,
                                                    so.scale, _stream()), "layernorm_split_f16")
        return so if split_out == "only" else (y, so)
    so = SplitT.empty(x.shape, x.device)
    _l.check(_l.load().aldm_layernorm_split(x.data_ptr(), _p(y), so.data_ptr(), M, Cc, gamma.data_ptr(),
                                            beta.data_ptr(), eps, so.parts, _stream()), "layernorm_split")
    return so if split_out == "only" else (y, so)


def _rowview(t: torch.Tensor, name: str):
    """[B, L, D] view (possibly a column slice of a wider contiguous buffer) -> (ptr, L, row pitch)."""
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.stride(2) == 1
            and t.stride(0) == t.shape[1] * t.stride(1)):
        raise RuntimeError(f"{name}: need [B, L, D] fp32 CUDA view with unit inner stride")
    return t.data_ptr(), t.shape[1], t.stride(1)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, *,
              mask: Optional[torch.Tensor] = None, scale: Optional[float] = None,
              split_out: Optional[str] = None):
    """softmax(scale * q k^T [mask]) v per head, head dim 32.  q: [B, Lq, heads*32] views.  split_out = "only": the
    result as a SplitT (pre-split operand of the out projection), "also": (fp32, SplitT)."""
    qp, Lq, ldq = _rowview(q, "attn.q")
    kp, Lk, ldk = _rowview(k, "attn.k")
    vp, Lv, ldv = _rowview(v, "attn.v")
    assert Lk == Lv and q.shape[2] == heads * 32
    B = q.shape[0]
    if scale is None:
        scale = 32 ** -0.5
    out = None if split_out == "only" else torch.empty((B, Lq, heads * 32), device=q.device, dtype=torch.float32)
    if mask is not None:
        mask = mask.to(torch.float32).reshape(B, Lk).contiguous()
    ev = None
    if ATTN_PROFILE is not None:   # bench.py's attention roofline: 2 products of 2*Lq*Lk*32 flops per (sample, head)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if not split_out:
        _l.check(_l.load().aldm_attention_d32(qp, kp, vp, out.data_ptr(), B, heads, Lq, Lk, ldq, ldk, ldv,
                                              heads * 32, _p(mask), scale, _stream()), "attention_d32")
        res = out
    else:
        so = SplitT.empty((B, Lq, heads * 32), q.device)
        _l.check(_l.load().aldm_attention_d32_split(qp, kp, vp, _p(out), so.data_ptr(), so.parts, B, heads, Lq, Lk, ldq,
                                                    ldk, ldv, heads * 32, _p(mask), scale, _stream()), "attention_d32_split")
        res = so if split_out == "only" else (out, so)
    if ev is not None:
        ev[1].record()
        ATTN_PROFILE.append((B, heads, Lq, Lk, mask is not None, 4.0 * B * heads * Lq * Lk * 32, ev[0], ev[1]))
    return res


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """T5LayerNorm over the last dim: weight * x * rsqrt(mean(x^2) + eps)."""
    _chk(x, "rmsnorm.x")
    Cc = x.shape[-1]
    y = torch.empty_like(x)
    _l.check(_l.load().aldm_rmsnorm(x.data_ptr(), y.data_ptr(), x.numel() // Cc, Cc, weight.data_ptr(), eps, _stream()),
             "rmsnorm")
    return y


def softmax_rows_bias(x: torch.Tensor, bias: torch.Tensor, keymask: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """x: [B, heads, q_rows, N] scores; bias [heads, q_rows, N] added to every batch entry; keymask [B, N] (0 = padded key,
    weight 0)."""
    _chk(x, "softmax_bias.x"); _chk(bias, "softmax_bias.bias"); _chk(keymask, "softmax_bias.keymask")
    B, heads, q_rows, N = x.shape
    assert bias.shape == (heads, q_rows, N) and keymask.shape == (B, N)
    y = torch.empty_like(x)
    _l.check(_l.load().aldm_softmax_rows_bias(x.data_ptr(), y.data_ptr(), B, heads, q_rows, N, scale, bias.data_ptr(),
                                              keymask.data_ptr(), _stream()), "softmax_rows_bias")
    return y


def rel_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, emb_k: torch.Tensor, emb_v: torch.Tensor,
                  mask: torch.Tensor) -> torch.Tensor:
    """Windowed relative-position self-attention (VITS phoneme encoder, attentions.py:239-289).  q/k/v: [B, T, heads*d]
    views; emb_k / emb_v: [2*window+1, d]; mask [B, T] floats (1 = token)."""
    qp, T, ldq = _rowview(q, "rel_attn.q")
    kp, Tk, ldk = _rowview(k, "rel_attn.k")
    vp, Tv, ldv = _rowview(v, "rel_attn.v")
    B = q.shape[0]
    d = q.shape[2] // heads
    assert T == Tk == Tv and emb_k.shape == emb_v.shape and emb_k.shape[1] == d and emb_k.shape[0] % 2 == 1
    _chk(emb_k, "rel_attn.emb_k"); _chk(emb_v, "rel_attn.emb_v"); _chk(mask, "rel_attn.mask")
    assert mask.shape == (B, T)
    out = torch.empty((B, T, heads * d), device=q.device, dtype=torch.float32)
    _l.check(_l.load().aldm_rel_attention(qp, kp, vp, out.data_ptr(), B, heads, T, d, ldq, ldk, ldv, heads * d,
                                          emb_k.data_ptr(), emb_v.data_ptr(), emb_k.shape[0] // 2, mask.data_ptr(),
                                          _stream()), "rel_attention")
    return out


def rowscale_add(x: torch.Tensor, s: torch.Tensor, res: Optional[torch.Tensor] = None, divide: bool = False) -> torch.Tensor:
    """y[r, c] = x[r, c] * s[r] (+ res[r, c]); x: [..., C], s: one scale per row.  divide: y = x / max(s[r], 1e-12)."""
    _chk(x, "rowscale.x"); _chk(s, "rowscale.s")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    assert s.numel() == rows
    if res is not None:
        _chk(res, "rowscale.res")
        assert res.shape == x.shape
    y = torch.empty_like(x)
    _l.check(_l.load().aldm_rowscale_add(x.data_ptr(), s.data_ptr(), _p(res), y.data_ptr(), rows, Cc, 1 if divide else 0,
                                         _stream()), "rowscale_add")
    return y


def resample_sinc(x: torch.Tensor, kernel: torch.Tensor, down: int, up: int, width: int, out_len: int) -> torch.Tensor:
    """Polyphase windowed-sinc resampling (torchaudio.functional.resample): x [B, T], kernel [up, taps] -> [B, out_len]."""
    _chk(x, "resample.x"); _chk(kernel, "resample.kernel")
    B, T = x.shape
    assert kernel.shape[0] == up
    y = torch.empty((B, out_len), device=x.device, dtype=torch.float32)
    _l.check(_l.load().aldm_resample_sinc(x.data_ptr(), kernel.data_ptr(), y.data_ptr(), B, T, out_len, down, up,
                                          kernel.shape[1], width, _stream()), "resample_sinc")
    return y


def power_spec(spec: torch.Tensor, F: int, ld_out: int) -> torch.Tensor:
    """spec [M, >= 2F] rows of [re | im] -> [M, ld_out] with re^2 + im^2 in the first F columns, zeros after."""
    _chk(spec, "power_spec.spec")
    M = spec.numel() // spec.shape[-1]
    out = torch.empty((M, ld_out), device=spec.device, dtype=torch.float32)
    _l.check(_l.load().aldm_power_spec(spec.data_ptr(), out.data_ptr(), M, F, spec.shape[-1], ld_out, _stream()),
             "power_spec")
    return out


def col_affine(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """y[..., c] = x[..., c] * scale[c] + shift[c]"""
    _chk(x, "col_affine.x"); _chk(scale, "col_affine.scale"); _chk(shift, "col_affine.shift")
    Cc = x.shape[-1]
    assert scale.numel() == Cc and shift.numel() == Cc
    y = torch.empty_like(x)
    _l.check(_l.load().aldm_col_affine(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), x.numel() // Cc, Cc,
                                       _stream()), "col_affine")
    return y


def bicubic_patchify(x: torch.Tensor, S: int, p: int) -> torch.Tensor:
    """x [B, T, mel] -> [B, (S/p)^2, p*p]: HTSAT's reshape_wav2img + the im2col of its patch-embedding conv."""
    _chk(x, "bicubic_patchify.x")
    B, T, Fm = x.shape
    out = torch.empty((B, (S // p) ** 2, p * p), device=x.device, dtype=torch.float32)
    _l.check(_l.load().aldm_bicubic_patchify(x.data_ptr(), out.data_ptr(), B, T, Fm, S, p, _stream()), "bicubic_patchify")
    return out


def token_mean(x: torch.Tensor) -> torch.Tensor:
    """[B, L, C] -> [B, C]"""
    _chk(x, "token_mean.x")
    B, L, Cc = x.shape
    y = torch.empty((B, Cc), device=x.device, dtype=torch.float32)
    _l.check(_l.load().aldm_token_mean(x.data_ptr(), y.data_ptr(), B, L, Cc, _stream()), "token_mean")
    return y


def row_cosine(a: torch.Tensor, b: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """[M, C] x [M, C] -> [M] cosine similarities (F.cosine_similarity's eps clamp)."""
    _chk(a, "row_cosine.a"); _chk(b, "row_cosine.b")
    assert a.shape == b.shape and a.dim() == 2
    out = torch.empty((a.shape[0],), device=a.device, dtype=torch.float32)
    _l.check(_l.load().aldm_row_cosine(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0], a.shape[1], eps, _stream()),
             "row_cosine")
    return out


def softmax_rows(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    _chk(x, "softmax.x")
    N = x.shape[-1]
    y = torch.empty_like(x)
    _l.check(_l.load().aldm_softmax_rows(x.data_ptr(), y.data_ptr(), x.numel() // N, N, scale,
                                         _stream()), "softmax_rows")
    return y


def softmax_rows_masked(x: torch.Tensor, keymask: torch.Tensor, q_pos0: int, scale: float = 1.0) -> torch.Tensor:
    """x: [B, heads, q_rows, N] attention scores; keymask [B, N] (1 = key takes part); causal: query row i sees keys
    <= q_pos0 + i.  Excluded keys get weight 0 (GPT-2 blocks of the sequence generator)."""
    _chk(x, "softmax_masked.x")
    _chk(keymask, "softmax_masked.keymask")
    B, heads, q_rows, N = x.shape
    assert keymask.shape == (B, N)
    y = torch.empty_like(x)
    _l.check(_l.load().aldm_softmax_rows_masked(x.data_ptr(), y.data_ptr(), B, heads, q_rows, N, scale,
                                                keymask.data_ptr(), q_pos0, _stream()), "softmax_rows_masked")
    return y


# ---- single-position decode step of the GPT-2 sequence generator (aldm_decode_linear / aldm_decode_attention) ----------
DECODE_MAX_ROWS = 16


def decode_linear(x: torch.Tensor, w_kn: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
                  ln: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None, act: int = ACT_NONE,
                  res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = act(LayerNorm(x) @ w_kn + bias) + res for M <= 16 rows: x [M, K], w_kn [K, N] fp32 (transformers Conv1D's own
    layout), ln = (gamma, beta, eps) or None.  Exact fp32 FMA, deterministic; a weight stream, not a tile GEMM."""
    _chk(x, "decode_linear.x"); _chk(w_kn, "decode_linear.w")
    M, K = x.shape
    assert w_kn.dim() == 2 and w_kn.shape[0] == K and 1 <= M <= DECODE_MAX_ROWS
    N = w_kn.shape[1]
    if res is not None:
        _chk(res, "decode_linear.res")
        assert res.shape == (M, N)
    y = torch.empty((M, N), device=x.device, dtype=torch.float32)
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    # the kernel takes raw pointers: a CPU / non-fp32 / strided bias or LayerNorm vector would be a fault or garbage, not an error
    if bias is not None:
        _chk(bias, "decode_linear.bias")
        assert bias.numel() == N, f"decode_linear: bias has {bias.numel()} entries for N = {N}"
    for t, nm in ((g, "ln gamma"), (b, "ln beta")):
        if t is not None:
            _chk(t, "decode_linear." + nm)
            assert t.numel() == K, f"decode_linear: {nm} has {t.numel()} entries for K = {K}"
    assert (g is None) == (b is None), "decode_linear: LayerNorm gamma and beta come together"
    _l.check(_l.load().aldm_decode_linear(x.data_ptr(), K, M, K, w_kn.data_ptr(), N, _p(bias), _p(g), _p(b), float(eps), act,
                                          _p(res), N, y.data_ptr(), N, _stream()), "decode_linear")
    return y


def decode_attention(qkv: torch.Tensor, pos: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                     keymask: torch.Tensor, heads: int, scale: Optional[float] = None) -> torch.Tensor:
    """The new position's attention over the key/value cache: qkv [B, 3 * heads * 64] (q | k | v rows of the new position),
    pos: one-element int64 DEVICE tensor (its cache slot), caches [B * heads, n_tot, 64] (updated in place at slot pos),
    keymask [B, n_tot] with slot pos already switched on.  Returns [B, heads * 64], heads merged."""
    _chk(qkv, "decode_attention.qkv"); _chk(k_cache, "decode_attention.k_cache"); _chk(v_cache, "decode_attention.v_cache")
    _chk(keymask, "decode_attention.keymask")
    B = qkv.shape[0]
    E = heads * 64
    n_tot = keymask.shape[1]
    assert qkv.shape == (B, 3 * E) and keymask.shape[0] == B
    assert k_cache.shape == (B * heads, n_tot, 64) and v_cache.shape == k_cache.shape
    assert pos.is_cuda and pos.dtype == torch.int64 and pos.numel() == 1
    out = torch.empty((B, E), device=qkv.device, dtype=torch.float32)
    _l.check(_l.load().aldm_decode_attention(qkv.data_ptr(), 3 * E, pos.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                                             keymask.data_ptr(), B, heads, n_tot, 0.125 if scale is None else float(scale),
                                             out.data_ptr(), E, _stream()), "decode_attention")
    return out


def decode_gemv(x: torch.Tensor, w_kn: torch.Tensor, *, xbias: Optional[torch.Tensor] = None, xact: int = ACT_NONE) -> torch.Tensor:
    """Split-K form of the decode GEMV (aldm_decode_gemv): x is [M, K] rows, or [P, M, K] partial slabs of a previous decode_gemv
    whose sum (+ xbias, through xact) is the operand.  Returns the partial slabs [S, M, N] of x_eff @ w_kn: their sum in slab
    order (decode_reduce_ln, decode_attention_parts or the next decode_gemv adds them) is the product."""
    _chk(x, "decode_gemv.x"); _chk(w_kn, "decode_gemv.w")
    P = 1 if x.dim() == 2 else x.shape[0]
    M, K = x.shape[-2], x.shape[-1]
    assert w_kn.dim() == 2 and w_kn.shape[0] == K and 1 <= M <= DECODE_MAX_ROWS
    N = w_kn.shape[1]
    lib = _l.load()
    S = lib.aldm_decode_gemv_slices(K, N)
    assert S > 0, f"decode_gemv: K = {K} cannot be sliced"
    if xbias is not None:
        _chk(xbias, "decode_gemv.xbias")
        assert xbias.numel() == K
    y = torch.empty((S, M, N), device=x.device, dtype=torch.float32)
    _l.check(lib.aldm_decode_gemv(x.data_ptr(), K, P, M * K, _p(xbias), xact, M, K, w_kn.data_ptr(), N, y.data_ptr(), _stream()),
             "decode_gemv")
    return y


def decode_reduce_ln(part: Optional[torch.Tensor], *, bias: Optional[torch.Tensor] = None, bias_row: Optional[torch.Tensor] = None,
                     res: Optional[torch.Tensor] = None, ln: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None,
                     want_h: bool = True, xn_out: Optional[torch.Tensor] = None):
    """h = sum of the partial slabs part[S, M, N] (slab order) + bias (row *bias_row of a table when bias_row is given) + res; with
    ln = (gamma, beta, eps) also LayerNorm(h).  Returns h, (h, xn) or xn (want_h = False); xn_out: a second destination of the
    normalised rows ([M, N]-shaped view with unit inner stride: a token slot of the generator's output)."""
    src = part if part is not None else res
    M, N = src.shape[-2], src.shape[-1]
    if part is not None:
        _chk(part, "decode_reduce_ln.part")
        assert part.dim() == 3
    for t, nm in ((bias, "bias"), (res, "res")):
        if t is not None:
            _chk(t, "decode_reduce_ln." + nm)
    if res is not None:
        assert res.shape == (M, N)
    if bias_row is not None:
        assert bias_row.is_cuda and bias_row.dtype == torch.int64 and bias_row.numel() == 1 and bias is not None and bias.dim() == 2
    elif bias is not None:
        assert bias.numel() == N
    dev = src.device
    h = torch.empty((M, N), device=dev, dtype=torch.float32) if want_h else None
    g, b, eps = ln if ln is not None else (None, None, 0.0)
    xn = torch.empty((M, N), device=dev, dtype=torch.float32) if ln is not None else None
    ldn2 = 0
    if xn_out is not None:
        assert ln is not None and xn_out.is_cuda and xn_out.dtype == torch.float32 and xn_out.shape == (M, N) and xn_out.stride(1) == 1
        ldn2 = xn_out.stride(0)
    _l.check(_l.load().aldm_decode_reduce_ln(_p(part), 0 if part is None else part.shape[0], M * N, N, _p(bias), _p(bias_row),
                                             _p(res), N, M, N, _p(h), N, _p(g), _p(b), float(eps), _p(xn), N, _p(xn_out), ldn2,
                                             _stream()), "decode_reduce_ln")
    if ln is None:
        return h
    return (h, xn) if want_h else xn


def decode_attention_parts(qkv_part: torch.Tensor, qbias: Optional[torch.Tensor], pos: torch.Tensor, k_cache: torch.Tensor,
                           v_cache: torch.Tensor, keymask: torch.Tensor, heads: int, scale: Optional[float] = None) -> torch.Tensor:
    """decode_attention over a sliced c_attn: q | k | v rows = qbias + the sum of the partial slabs qkv_part [S, B, 3 E]."""
    _chk(qkv_part, "decode_attention_parts.qkv"); _chk(k_cache, "decode_attention_parts.k_cache")
    _chk(v_cache, "decode_attention_parts.v_cache"); _chk(keymask, "decode_attention_parts.keymask")
    S, B = qkv_part.shape[0], qkv_part.shape[1]
    E = heads * 64
    n_tot = keymask.shape[1]
    assert qkv_part.shape == (S, B, 3 * E) and keymask.shape[0] == B
    assert k_cache.shape == (B * heads, n_tot, 64) and v_cache.shape == k_cache.shape
    assert pos.is_cuda and pos.dtype == torch.int64 and pos.numel() == 1
    if qbias is not None:
        _chk(qbias, "decode_attention_parts.qbias")
        assert qbias.numel() == 3 * E
    out = torch.empty((B, E), device=qkv_part.device, dtype=torch.float32)
    _l.check(_l.load().aldm_decode_attention_parts(qkv_part.data_ptr(), 3 * E, S, B * 3 * E, _p(qbias), pos.data_ptr(),
                                                   k_cache.data_ptr(), v_cache.data_ptr(), keymask.data_ptr(), B, heads, n_tot,
                                                   0.125 if scale is None else float(scale), out.data_ptr(), E, _stream()),
             "decode_attention_parts")
    return out


def geglu(x: torch.Tensor) -> torch.Tensor:
    _chk(x, "geglu.x")
    C2 = x.shape[-1]
    M = x.numel() // C2
    y = torch.empty((*x.shape[:-1], C2 // 2), device=x.device, dtype=torch.float32)
    _l.check(_l.load().aldm_geglu(x.data_ptr(), y.data_ptr(), M, C2 // 2, _stream()), "geglu")
    return y


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    t = t.to(torch.float32).contiguous()
    out = torch.empty((t.shape[0], dim), device=t.device, dtype=torch.float32)
    _l.check(_l.load().aldm_timestep_embedding(t.data_ptr(), out.data_ptr(), t.shape[0], dim,
                                               max_period, _stream()), "timestep_embedding")
    return out


def nchw_to_nhwc(x: torch.Tensor, rep: int = 1) -> torch.Tensor:
    _chk(x, "nchw_to_nhwc.x")
    B, Cc, H, W = x.shape
    y = torch.empty((rep * B, H, W, Cc), device=x.device, dtype=torch.float32)
    _l.check(_l.load().aldm_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), B, Cc, H * W, rep, _stream()),
             "nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    _chk(x, "nhwc_to_nchw.x")
    B, H, W, Cc = x.shape
    y = torch.empty((B, Cc, H, W), device=x.device, dtype=torch.float32)
    _l.check(_l.load().aldm_nhwc_to_nchw(x.data_ptr(), y.data_ptr(), B, Cc, H * W, _stream()),
             "nhwc_to_nchw")
    return y


def ddim_step(x: torch.Tensor, eps: torch.Tensor, noise: torch.Tensor, coef: torch.Tensor,
              x_prev: Optional[torch.Tensor] = None, pred_x0: Optional[torch.Tensor] = None):
    """Fused CFG combine + DDIM update.  eps: [2, *x.shape] (uncond, cond) or [*x.shape]."""
    for t, n in ((x, "x"), (eps, "eps"), (noise, "noise"), (coef, "coef")):
        _chk(t, "ddim." + n)
    if x_prev is None:
        x_prev = torch.empty_like(x)
    if pred_x0 is None:
        pred_x0 = torch.empty_like(x)
    _l.check(_l.load().aldm_ddim_step(x.data_ptr(), eps.data_ptr(), noise.data_ptr(), coef.data_ptr(),
                                      x_prev.data_ptr(), pred_x0.data_ptr(), x.numel(), _stream()),
             "ddim_step")
    return x_prev, pred_x0


def ddim_step_indexed(x: torch.Tensor, eps: torch.Tensor, noise_tab: torch.Tensor, coef_tab: torch.Tensor,
                      step_idx: torch.Tensor, pred_x0: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ops.ddim_step with the step's coefficient / noise rows selected on the device by the int32 counter `step_idx`, in
    place on x (aldm_ddim_step_indexed).  noise_tab: [S, *x.shape], coef_tab: [S, >= 7]."""
    for t, n in ((x, "x"), (eps, "eps"), (noise_tab, "noise_tab"), (coef_tab, "coef_tab")):
        _chk(t, "ddim_indexed." + n)
    assert step_idx.dtype == torch.int32 and step_idx.is_cuda and noise_tab.shape[1:] == x.shape and coef_tab.dim() == 2
    assert noise_tab.shape[0] == coef_tab.shape[0]
    _l.check(_l.load().aldm_ddim_step_indexed(x.data_ptr(), eps.data_ptr(), noise_tab.data_ptr(), coef_tab.data_ptr(),
                                              step_idx.data_ptr(), _p(pred_x0), x.numel(), coef_tab.shape[1], _stream()),
             "ddim_step_indexed")
    return x


def step_advance(step_idx: torch.Tensor, t_tab: torch.Tensor, t_cur: torch.Tensor) -> None:
    """step_idx += 1; t_cur = t_tab[min(step_idx, S - 1)] on the device (aldm_step_advance)."""
    _chk(t_tab, "step_advance.t_tab"); _chk(t_cur, "step_advance.t_cur")
    assert step_idx.dtype == torch.int32 and t_tab.dim() == 2 and t_tab.shape[1] == t_cur.numel()
    _l.check(_l.load().aldm_step_advance(step_idx.data_ptr(), t_tab.data_ptr(), t_cur.data_ptr(), t_tab.shape[1],
                                         t_tab.shape[0], _stream()), "step_advance")


def ddpm_step(x: torch.Tensor, eps: torch.Tensor, noise: torch.Tensor, coef: torch.Tensor,
              x_prev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Ancestral DDPM update (ddpm.py:357-373, 1127-1181); coef: device [>=5] row, see aldm_hip.h."""
    for t, n in ((x, "x"), (eps, "eps"), (noise, "noise"), (coef, "coef")):
        _chk(t, "ddpm." + n)
    if x_prev is None:
        x_prev = torch.empty_like(x)
    _l.check(_l.load().aldm_ddpm_step(x.data_ptr(), eps.data_ptr(), noise.data_ptr(), coef.data_ptr(),
                                      x_prev.data_ptr(), x.numel(), _stream()), "ddpm_step")
    return x_prev


def inpaint_blend(x: torch.Tensor, x0: torch.Tensor, qnoise: torch.Tensor, mask: torch.Tensor,
                  coef: torch.Tensor) -> torch.Tensor:
    """In place: x = q_sample(x0, t; qnoise)*mask + (1-mask)*x (ddim.py:226-231); coef = {sa, so}."""
    for t, n in ((x, "x"), (x0, "x0"), (qnoise, "qnoise"), (mask, "mask"), (coef, "coef")):
        _chk(t, "inpaint." + n)
    assert x.shape == x0.shape == qnoise.shape == mask.shape
    _l.check(_l.load().aldm_inpaint_blend(x0.data_ptr(), qnoise.data_ptr(), mask.data_ptr(),
                                          coef.data_ptr(), x.data_ptr(), x.numel(), _stream()),
             "inpaint_blend")
    return x


def axpby(a: torch.Tensor, b: Optional[torch.Tensor], alpha: float, beta: float = 0.0,
          out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(a, "axpby.a")
    if out is None:
        out = torch.empty_like(a)
    _l.check(_l.load().aldm_axpby(a.data_ptr(), _p(b), out.data_ptr(), alpha, beta, a.numel(),
                                  _stream()), "axpby")
    return out


def reflect_pad_1d(x: torch.Tensor, pad: int) -> torch.Tensor:
    """[B, T] -> [B, pitch] rows holding the T + 2*pad reflect-padded samples (stft.py:60-64);
    pitch = round_up(T + 2*pad, 4) so every frame start stays 16-byte aligned."""
    _chk(x, "reflect_pad.x")
    B, T = x.shape
    pitch = (T + 2 * pad + 3) // 4 * 4
    buf = torch.zeros((B, pitch), device=x.device, dtype=torch.float32)
    _l.check(_l.load().aldm_reflect_pad_1d(x.data_ptr(), buf.data_ptr(), B, T, pad, pitch, _stream()),
             "reflect_pad_1d")
    return buf


def frames_gemm(sig: torch.Tensor, frames: int, hop: int, pw: Packed) -> torch.Tensor:
    """STFT as an implicit GEMM (stft.py:67-72): out[b, f, n] = sum_k sig[b, f*hop + k] * W[k, n].
    Rows of A overlap (pixel pitch = hop < K = n_fft); one batch (grid z) per signal."""
    _chk(sig, "frames_gemm.sig")
    B, pitch = sig.shape
    K = pw.K
    assert K % 4 == 0 and hop % 4 == 0 and (frames - 1) * hop + K <= pitch
    out = torch.empty((B, frames, pw.N), device=sig.device, dtype=torch.float32)
    d = IgemmDesc()
    d.x1 = sig.data_ptr(); d.C1 = K; d.pix1 = hop; d.B = 1; d.H = 1; d.W = frames
    d.up_h = d.up_w = 1
    d.KH = d.KW = d.SH = d.SW = d.DH = d.DW = 1
    d.OH = 1; d.OW = frames
    d.w = pw.data.data_ptr(); d.b_mode = B_PACKED; d.K = K; d.N = pw.N
    d.out = out.data_ptr(); d.ldo = pw.N; d.alpha = 1.0
    d.batch = B; d.stride_x = pitch; d.stride_w = 0; d.stride_o = frames * pw.N
    _igemm(d, "igemm(frames)")
    return out


def mag_phase(spec: torch.Tensor, F: int, ld_mag: int, want_phase: bool = True):
    _chk(spec, "mag_phase.spec")
    M = spec.numel() // spec.shape[-1]
    mag = torch.empty((M, ld_mag), device=spec.device, dtype=torch.float32)
    phase = torch.empty((M, F), device=spec.device, dtype=torch.float32) if want_phase else None
    _l.check(_l.load().aldm_mag_phase(spec.data_ptr(), mag.data_ptr(), _p(phase), M, F,
                                      spec.shape[-1], ld_mag, _stream()), "mag_phase")
    return mag, phase


def row_l2norm(x: torch.Tensor, F: int) -> torch.Tensor:
    """x: [M, ld] -> [M] with out[m] = ||x[m, :F]||_2."""
    _chk(x, "row_l2norm.x")
    M, ld = x.shape
    out = torch.empty((M,), device=x.device, dtype=torch.float32)
    _l.check(_l.load().aldm_row_l2norm(x.data_ptr(), out.data_ptr(), M, F, ld, _stream()), "row_l2norm")
    return out
