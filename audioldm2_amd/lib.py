"""ctypes binding of libaldm_hip.so (the C ABI declared in include/aldm_hip.h).

The library is the product: there is NO CPU / PyTorch fallback behind these bindings.  If the
shared object is missing or a call fails, a RuntimeError is raised (the reference's convention is
plain Python exceptions, e.g. openaimodel.py:858-860).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ALDM_LIB_PATH") or os.path.join(_HERE, "libaldm_hip.so")  # override: debug builds
ABI_VERSION = 9

ACT_NONE, ACT_SILU, ACT_LRELU, ACT_TANH, ACT_LOGCLAMP, ACT_GELU, ACT_GELU_TANH = range(7)
B_PACKED, B_NT = 0, 1
EPI_PLAIN, EPI_GEGLU, EPI_QKV = 0, 1, 2
FMT_BF16, FMT_F16 = 0, 1


class IgemmDesc(C.Structure):
    """Mirror of `struct aldm_igemm_desc` (include/aldm_hip.h) — keep field order identical."""

    _fields_ = [
        ("x1", C.c_void_p), ("x2", C.c_void_p),
        ("C1", C.c_int32), ("C2", C.c_int32),
        ("pix1", C.c_int32), ("pix2", C.c_int32),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("up_h", C.c_int32), ("up_w", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("SH", C.c_int32), ("SW", C.c_int32),
        ("PH", C.c_int32), ("PW", C.c_int32), ("DH", C.c_int32), ("DW", C.c_int32),
        ("OH", C.c_int32), ("OW", C.c_int32),
        ("pre_scale", C.c_void_p), ("pre_shift", C.c_void_p),
        ("pre_act", C.c_int32), ("pre_slope", C.c_float),
        ("w", C.c_void_p),
        ("b_mode", C.c_int32), ("ldb", C.c_int32),
        ("K", C.c_int32), ("N", C.c_int32),
        ("bias", C.c_void_p), ("rowbias", C.c_void_p), ("res", C.c_void_p), ("out", C.c_void_p),
        ("ldo", C.c_int32), ("act", C.c_int32), ("act_slope", C.c_float), ("alpha", C.c_float),
        ("accumulate", C.c_int32),
        ("out_mul", C.c_int32), ("out_off", C.c_int32), ("out_len", C.c_int32),
        ("batch", C.c_int32),
        ("stride_x", C.c_int64), ("stride_w", C.c_int64), ("stride_o", C.c_int64),
        ("rowbias_ld", C.c_int32), ("epi_mode", C.c_int32),
        ("ws", C.c_void_p), ("ws_floats", C.c_int64),
        ("hint_bm", C.c_int32), ("hint_bn", C.c_int32), ("hint_splits", C.c_int32), ("hint_kgroups", C.c_int32),
        ("w_split", C.c_void_p), ("hint_mma", C.c_int32), ("hint_stages", C.c_int32),
        ("a_split", C.c_void_p), ("out_split", C.c_void_p), ("out_split_c", C.c_int32), ("split_parts", C.c_int32),
        ("out_split_act", C.c_int32), ("out_split_slope", C.c_float),
        ("k_split", C.c_void_p), ("vt_split", C.c_void_p), ("qkv_c", C.c_int32), ("qkv_rows", C.c_int32),
        ("a_fmt", C.c_int32), ("acc_scale", C.c_float), ("out_split_parts", C.c_int32),
        ("out_split_fmt", C.c_int32), ("out_split_scale", C.c_float), ("vt_scale", C.c_float),
    ]


_SIGS = {
    "aldm_version": (C.c_int, []),
    "aldm_last_error": (C.c_char_p, []),
    "aldm_igemm": (C.c_int, [C.POINTER(IgemmDesc), C.c_void_p]),
    "aldm_igemm_plan": (C.c_int, [C.POINTER(IgemmDesc), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(C.c_int)]),
    "aldm_igemm_ws_floats": (C.c_int64, [C.POINTER(IgemmDesc)]),
    "aldm_igemm_plan_stages": (C.c_int, [C.POINTER(IgemmDesc)]),
    "aldm_igemm_force": (None, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "aldm_igemm_force_stages": (None, [C.c_int]),
    "aldm_igemm_wave8_mask": (C.c_int, [C.c_int]),
    "aldm_split_image_bytes": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "aldm_split_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_split_rows_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_layernorm_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_float, C.c_int, C.c_void_p]),
    "aldm_attention_d32_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_float, C.c_void_p]),
    "aldm_split_bytes_parts": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
    "aldm_pack_split_bf16_parts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "aldm_pack_split_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "aldm_split_rows_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "aldm_groupnorm_split_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "aldm_layernorm_split_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_float, C.c_float, C.c_void_p]),
    "aldm_igemm_mma": (C.c_int, [C.c_int]),
    "aldm_split_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "aldm_pack_split_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "aldm_pack_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "aldm_pack_kn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_int64, C.c_int64, C.c_void_p]),
    "aldm_groupnorm_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "aldm_gn_ws_floats": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "aldm_groupnorm_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_float, C.c_void_p]),
    "aldm_attention_mma": (C.c_int, [C.c_int]),
    "aldm_attention_sched": (C.c_int, [C.c_int]),
    "aldm_attention_d32_presplit_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                                  C.c_float, C.c_float, C.c_void_p]),
    "aldm_attention_d32_presplit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "aldm_attention_d32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_float, C.c_void_p]),
    "aldm_rel_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_void_p]),
    "aldm_rowscale_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "aldm_rmsnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p]),
    "aldm_softmax_rows_bias": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "aldm_resample_sinc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p]),
    "aldm_power_spec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "aldm_col_affine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "aldm_bicubic_patchify": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "aldm_token_mean": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "aldm_row_cosine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "aldm_decode_linear": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_decode_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                        C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_decode_gemv_slices": (C.c_int, [C.c_int, C.c_int]),
    "aldm_decode_gemv": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_int, C.c_void_p, C.c_void_p]),
    "aldm_decode_reduce_ln": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                        C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_decode_attention_parts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                              C.c_void_p]),
    "aldm_softmax_rows_masked": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                           C.c_void_p, C.c_int, C.c_void_p]),
    "aldm_softmax_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float,
                                    C.c_void_p]),
    "aldm_geglu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "aldm_timestep_embedding": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                          C.c_void_p]),
    "aldm_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p]),
    "aldm_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "aldm_ddim_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int64, C.c_void_p]),
    "aldm_ddim_step_indexed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_int, C.c_void_p]),
    "aldm_step_advance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "aldm_ddpm_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                 C.c_void_p]),
    "aldm_inpaint_blend": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int64, C.c_void_p]),
    "aldm_axpby": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int64,
                             C.c_void_p]),
    "aldm_reflect_pad_1d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    "aldm_mag_phase": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                 C.c_int, C.c_void_p]),
    "aldm_row_l2norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
}

# every symbol include/aldm_hip.h declares (checked by tests/test_abi.py without a GPU)
EXPORTED_SYMBOLS = tuple(_SIGS.keys())
# ... and what only the -DALDM_TEST_HOOKS variant (libaldm_hip_testhooks.so, loaded through $ALDM_LIB_PATH by one test) adds
TEST_HOOK_SIGS = {"aldm_debug_drop_product": (C.c_int, [C.c_int])}
TESTHOOKS_LIB_PATH = os.path.join(_HERE, "libaldm_hip_testhooks.so")

_lib = None


def source_hash(family: str = "all") -> str:
    """16 hex digits over the kernel sources and the ABI header: profile artefacts (profiles/*.json) carry it so a
    reader — and bench.py — can tell whether they describe the code that is running.  family = "igemm": only what the igemm
    translation units are built from (igemm*.hip / igemm*.h, common.h, the ABI header) — the stamp of per-launch counters of
    igemm kernels, which a change to another kernel family (attention, norms, decode) does not invalidate."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   glob.glob(os.path.join(os.path.dirname(_HERE), "include", "*.h")))
    if family == "igemm":
        files = [f for f in files if os.path.basename(f).startswith("igemm") or os.path.basename(f) in ("common.h", "aldm_hip.h")]
    elif family != "all":
        raise ValueError(f"source_hash: unknown family {family!r}")
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def tuning_hash() -> str:
    """16 hex digits over the shipped geometry tables (audioldm2_amd/tuning/*.json).  Per-launch counter records carry it next to
    source_hash: an instantiation's launch set — hence its AVERAGE bytes per launch — follows the table that routes geometries to it."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_HERE, "tuning", "*.json"))):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load():
    """Load libaldm_hip.so once; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP kernel library has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C audioldm2_amd/csrc`). "
            "There is no CPU fallback for the sampling hot path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in TEST_HOOK_SIGS.items():   # bound only when the loaded library is the test-hook variant
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    ver = lib.aldm_version()
    if ver != ABI_VERSION:
        raise RuntimeError(f"libaldm_hip.so ABI version {ver} != expected {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().aldm_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libaldm_hip {what} failed ({rc}): {msg}")
