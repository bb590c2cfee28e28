"""CPU tests of the drop-in boundary: libaldm_hip.so loads and exports every symbol that
include/aldm_hip.h declares; the ctypes struct mirrors the C struct; errors surface as
RuntimeError; the product has no CPU fallback.  (No kernel is launched here.)"""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "aldm_hip.h")


def declared_functions(test_hooks=False):
    """Entry points the header declares for the release library (test_hooks: only those under #ifdef ALDM_TEST_HOOKS)."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    hooks = "".join(re.findall(r"#ifdef ALDM_TEST_HOOKS(.*?)#endif", src, flags=re.S))
    if test_hooks:
        src = hooks
    else:
        src = re.sub(r"#ifdef ALDM_TEST_HOOKS.*?#endif", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aldm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from audioldm2_amd import lib
    l = lib.load()
    names = declared_functions()
    assert len(names) >= 19
    for n in names:
        assert hasattr(l, n), f"{n} declared in include/aldm_hip.h but not exported"
    assert sorted(lib.EXPORTED_SYMBOLS) == names, "lib.py signature table out of sync with the header"
    assert l.aldm_version() == lib.ABI_VERSION


def test_test_hooks_live_in_the_variant_library_only():
    """the precision-breaking switch is not in the shipped library.  libaldm_hip_testhooks.so (same sources,
    -DALDM_TEST_HOOKS) exports everything the release library does plus the hooks; the release library exports none of them."""
    from audioldm2_amd import lib
    hooks = declared_functions(test_hooks=True)
    assert hooks == sorted(lib.TEST_HOOK_SIGS) == ["aldm_debug_drop_product"]
    rel = lib.load()
    var = ctypes.CDLL(lib.TESTHOOKS_LIB_PATH)
    for n in hooks:
        assert not hasattr(rel, n), f"{n} must not be exported by libaldm_hip.so"
        assert hasattr(var, n)
    for n in declared_functions():
        assert hasattr(var, n)
    assert var.aldm_version() == lib.ABI_VERSION


def test_igemm_desc_layout_matches_c_struct(tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
    from audioldm2_amd.lib import IgemmDesc
    fields = [f[0] for f in IgemmDesc._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(){",
            'printf("%zu\\n", sizeof(aldm_igemm_desc));']
    for f in fields:
        prog.append(f'printf("%zu\\n", offsetof(aldm_igemm_desc, {f}));')
    prog.append("return 0;}")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", str(c), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).decode().split()]
    assert vals[0] == ctypes.sizeof(IgemmDesc)
    for f, off in zip(fields, vals[1:]):
        assert getattr(IgemmDesc, f).offset == off, f


def test_argument_validation_reports_errors_without_gpu():
    """Validation happens before any launch, so it is observable on a CPU-only box."""
    from audioldm2_amd import lib
    l = lib.load()
    d = lib.IgemmDesc()
    rc = l.aldm_igemm(ctypes.byref(d), None)
    assert rc != 0
    with pytest.raises(RuntimeError, match="null x1/w/out"):
        lib.check(rc, "igemm")
    assert l.aldm_groupnorm_stats(None, None, 1, 1, 8, 0, 32, 1e-5, None, None, None, None, None, None) != 0


def test_decode_entry_points_validate_before_launch():
    """aldm_decode_linear / aldm_decode_attention (ABI v7) refuse what their kernels cannot run — more than 16 rows, a K the
    passes do not tile, a fused LayerNorm over rows longer than the registers hold, a cache longer than GPT-2's 1024
    positions — before any launch, so it is observable without a GPU."""
    from audioldm2_amd import lib
    l = lib.load()
    p = ctypes.c_void_p(4096)   # never dereferenced: validation fails first

    def lin(M, K, N, ln=False):
        g = p if ln else None
        return l.aldm_decode_linear(p, K, M, K, p, N, None, g, g, 1e-5, 0, None, 0, p, N, None)
    for M, K, N, ln, msg in ((17, 768, 768, False, "rows"), (0, 768, 768, False, "rows"), (8, 48, 64, False, "multiple of 64"),
                             (8, 1024 + 64, 64, False, "multiple of 128"), (8, 2048, 64, True, "LayerNorm")):
        assert lin(M, K, N, ln) != 0
        assert msg in l.aldm_last_error().decode()
    assert l.aldm_decode_attention(p, 3 * 768, p, p, p, p, 8, 12, 1025, 0.125, p, 768, None) != 0
    assert "1024" in l.aldm_last_error().decode()
    assert l.aldm_decode_attention(p, 768, p, p, p, p, 8, 12, 64, 0.125, p, 768, None) != 0
    assert "pitch" in l.aldm_last_error().decode()


def test_every_tuned_dma_table_entry_plans_onto_the_hinted_kernel():
    """The shipped tuning tables of the DMA-fed kernel family (audioldm2_amd/tuning/mi355x_igemm_dma*.json): every entry's
    (tile, split-K, kernel form + ring depth) hint is ACCEPTED by the host-side planner for the geometry it is keyed on — a
    stale entry (a form that cannot run that shape any more) would silently fall back to the cost model and lose its tuning."""
    import json
    from audioldm2_amd import lib, ops
    l = lib.load()
    n = 0
    for name, parts in (("mi355x_igemm_dma.json", 3), ("mi355x_igemm_dma_bf16x3.json", 2)):
        with open(os.path.join(ROOT, "audioldm2_amd", "tuning", name)) as f:
            entries = json.load(f)["entries"]
        assert len(entries) > 100
        for key, v in entries.items():
            fields = key.split(",")
            d = lib.IgemmDesc()
            for k, x in zip(ops._TUNE_FIELDS, fields):
                setattr(d, k, int(x))
            d.K = (d.C1 + d.C2) * d.KH * d.KW
            d.a_split, d.split_parts, d.w, d.w_split = 4096, parts, 4096, 8192   # never dereferenced by the planner
            d.out, d.alpha = 1 << 20, 1.0
            d.ldo = d.N // 2 if d.epi_mode == lib.EPI_GEGLU else d.N
            d.ws, d.ws_floats = 16, 1 << 40
            d.hint_bm, d.hint_bn, d.hint_splits, d.hint_stages = v[:4]
            bm, bn, sp = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            assert l.aldm_igemm_plan(ctypes.byref(d), ctypes.byref(bm), ctypes.byref(bn), None, ctypes.byref(sp), None,
                                     None) == 0, (key, l.aldm_last_error())
            assert (bm.value, bn.value, sp.value) == tuple(v[:3]), (name, key, v)
            stages = l.aldm_igemm_plan_stages(ctypes.byref(d))
            assert stages == v[3] or (v[3] % 100 == 0 and stages // 100 == v[3] // 100), (name, key, v, stages)
            n += 1
    assert n >= 300


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from audioldm2_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.load()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under audioldm2_amd/ may import it."""
    pkg = os.path.join(ROOT, "audioldm2_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn


def test_hot_modules_refuse_cpu_tensors():
    import torch
    from audioldm2_amd.unet import UNetModel
    from oracle import cases
    m = UNetModel(**cases.UNET_TINY)
    x, t, ctxs, masks, y = cases.unet_inputs(cases.UNET_TINY, 1, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(x, t, context_list=ctxs, context_attn_mask_list=masks)


def _plan(l, lib, d):
    bm, bn, sp, kg, mma = (ctypes.c_int() for _ in range(5))
    fl = ctypes.c_int64()
    rc = l.aldm_igemm_plan(ctypes.byref(d), ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(fl), ctypes.byref(sp),
                           ctypes.byref(kg), ctypes.byref(mma))
    lib.check(rc, "plan")
    return bm.value, bn.value, sp.value, kg.value, mma.value, fl.value


def _conv_desc(lib, B=16, H=256, W=16, C=128, N=128, k=3):
    d = lib.IgemmDesc()
    d.x1, d.w, d.out = 0x1000, 0x2000, 0x3000  # never dereferenced: the plan query is host-only
    d.C1, d.B, d.H, d.W = C, B, H, W
    d.KH = d.KW = k
    d.SH = d.SW = d.DH = d.DW = 1
    d.PH = d.PW = k // 2
    d.OH, d.OW = H, W
    d.K, d.N, d.ldo = k * k * C, N, N
    d.b_mode = lib.B_PACKED
    d.batch = 1
    return d


def test_plan_selects_the_matrix_core_path_on_the_host():
    """aldm_igemm_plan (host-only query): a descriptor with a split weight image plans onto the bf16-split kernels,
    without one (or with the fp32 override / an activation x activation product / a hint) onto the fp32 MFMA; the
    FLOP count is the algorithmic 2*M*N*K either way."""
    from audioldm2_amd import lib
    l = lib.load()
    d = _conv_desc(lib)
    bm, bn, sp, kg, mma, fl = _plan(l, lib, d)
    assert mma == 0 and fl == 2 * (16 * 256 * 16) * 128 * (9 * 128) and (bm, bn) in {(128, 128), (64, 128), (128, 64), (64, 64)}
    d.w_split = 0x4000
    assert _plan(l, lib, d)[4] == 1
    d.hint_mma = 1  # tuned table says: fp32 MFMA for this shape
    assert _plan(l, lib, d)[4] == 0
    d.hint_mma = 0
    prev = l.aldm_igemm_mma(1)  # thread-local override: fp32 MFMA always
    try:
        assert _plan(l, lib, d)[4] == 0
    finally:
        l.aldm_igemm_mma(prev)
    assert _plan(l, lib, d)[4] == 1
    d.b_mode, d.ldb = lib.B_NT, d.K  # activation x activation: no split image can exist
    assert _plan(l, lib, d)[4] == 0
    d.b_mode, d.ldb = lib.B_PACKED, 0
    d.N, d.ldo = 8, 8  # N <= 32 -> the 128x32 tile, which has no bf16-split variant
    assert _plan(l, lib, d)[:2] == (128, 32) and _plan(l, lib, d)[4] == 0


def test_split_image_size_and_forced_tiles():
    from audioldm2_amd import lib
    l = lib.load()
    assert l.aldm_split_bytes(72, 40) == 12 * 3 * 64 * 16  # 4*ceil(72/32) k-octets x 3 parts x Npad x 16 B
    assert l.aldm_split_bytes(1152, 128) == 144 * 3 * 128 * 16 == 1152 * 128 * 6
    assert l.aldm_split_bytes(0, 5) == 0
    d = _conv_desc(lib)
    d.w_split = 0x4000
    l.aldm_igemm_force(64, 64, 1, 2)
    try:
        assert _plan(l, lib, d)[:5] == (64, 64, 1, 2, 1)
        l.aldm_igemm_force(256, 128, 1, 1)
        assert l.aldm_igemm_plan(ctypes.byref(d), None, None, None, None, None, None) != 0
        with pytest.raises(RuntimeError, match="unsupported forced/hinted tile"):
            lib.check(1, "plan")
    finally:
        l.aldm_igemm_force(0, 0, 0, 0)
    assert l.aldm_igemm_wave8_mask(5) == 5 and l.aldm_igemm_wave8_mask(-1) in (1, int(os.environ.get("ALDM_IGEMM_W8", "1")))
