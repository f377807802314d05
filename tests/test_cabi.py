"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/mi355_flow.h
declares (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from mi355_flow import _lib


def _declared():
    src = open(os.path.join(ROOT, "include", "mi355_flow.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mi355_flow.h but not exported"
    assert sorted(_lib.SIGNATURES) == names  # the ctypes table binds exactly the declared ABI
    assert lib.mi355_version() == 2


def test_error_convention_without_gpu():
    lib = _lib.load()
    cfg = _lib.ModelCfg(16, 16, 2, 2, 2, 32, 128, 128, 16, 256, 4, 1, 1e-6)  # head_dim 32: rejected before any HIP call
    h = C.c_void_p()
    assert lib.mi355_engine_create(C.byref(cfg), C.byref(h)) != 0
    assert b"head_dim" in lib.mi355_last_error()
    with pytest.raises(RuntimeError, match="head_dim"):
        _lib.check(lib.mi355_engine_create(C.byref(cfg), C.byref(h)), "engine_create")
    assert lib.mi355_engine_destroy(None) == 0 and lib.mi355_plan_destroy(None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU / PyTorch fallback"):
        _lib.load()
