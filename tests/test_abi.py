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


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
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
