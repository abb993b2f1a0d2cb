"""CPU: the C-ABI library loads and exports every symbol that include/*.h declares (no compute
calls: there is no GPU here), and the product package never reaches into oracle/."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not h.endswith(".h"):
            continue
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(hhgt_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from haplohyped_varawareml_amd import build
    lib = build.build()
    L = ctypes.CDLL(lib)
    decl = declared_functions()
    assert len(decl) >= 15
    missing = [n for n in decl if not hasattr(L, n)]
    assert not missing, missing
    L.hhgt_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.hhgt_version()


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from haplohyped_varawareml_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    rc = L.hhgt_ctx_create(0, ctypes.byref(h))
    assert rc == -6 and b"no CPU path" in L.hhgt_last_error()
    from haplohyped_varawareml_amd.device import Context
    with pytest.raises(_lib.HhgtError):
        Context(0)


def test_layout_helpers_are_host_side():
    from haplohyped_varawareml_amd import _lib, device
    L = _lib.load()
    lay = device.make_layout(2504, 100000, sc=64, vc=16384)
    assert lay.v_capacity == 7 * 16384
    assert device.layout_bytes(lay) == 40 * 64 * 7 * 16384 * 2
    off = L.hhgt_layout_offset(ctypes.byref(lay), 65, 16385)
    assert off == (((1 * 40 + 1) * 64 + 1) * 16384 + 1) * 2
    d = device.make_layout(2504, 100000)             # default geometry: 64 x 8192 chunks
    assert d.vc == 8192 and d.v_capacity == 13 * 8192
    dense = device.make_layout(3, 1000, sc=0, vc=0)
    assert dense.v_capacity == 1024 and L.hhgt_layout_offset(ctypes.byref(dense), 2, 5) == (2 * 1024 + 5) * 2


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "haplohyped_varawareml_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "liboracle" not in src and not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
