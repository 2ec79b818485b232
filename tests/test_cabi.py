"""CPU checks of the C-ABI boundary: the library loads, exports every symbol that
include/sqfa_hip.h declares, and validates arguments on the host (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from sqfa_amd import _lib

HEADER = os.path.join(ROOT, "include", "sqfa_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sqfa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 9
    for name in names:
        assert hasattr(lib, name), f"{name} declared in sqfa_hip.h but not exported"
    # and the python loader binds exactly that set
    assert sorted(_lib.PROTOTYPES) == names


def test_identification():
    lib = _lib.load()
    assert lib.sqfa_hip_arch() == b"gfx950"
    assert lib.sqfa_hip_version() >= 1000
    assert lib.sqfa_hip_max_dim() >= 64


@pytest.mark.parametrize("dtype", [_lib.SQFA_F32, _lib.SQFA_F64])
@pytest.mark.parametrize("m", [1, 2, 4, 5, 8, 9, 16, 17, 32, 33, 40, 48, 64])
def test_tiling_and_workspace(m, dtype):
    lib = _lib.load()
    out = [ctypes.c_int() for _ in range(5)]
    assert lib.sqfa_airm_tiling(1000, 0, m, dtype, *[ctypes.byref(v) for v in out]) == 0
    ti, tj, nbi, nbj, mr = [v.value for v in out]
    assert mr >= m and ti * nbi >= 1000 and tj * nbj >= 1000 and 64 % ti == 0
    nbytes = lib.sqfa_airm_workspace_bytes(1000, 0, m, dtype)
    esz = 4 if dtype == _lib.SQFA_F32 else 8
    assert nbytes >= 2 * 1000 * mr * mr * esz
    # cross mode has its own geometry
    assert lib.sqfa_airm_workspace_bytes(10, 7, m, dtype) > 0


def test_argument_validation_on_the_host():
    lib = _lib.load()
    assert lib.sqfa_airm_workspace_bytes(10, 0, 1000, _lib.SQFA_F32) == 0          # unsupported m
    assert lib.sqfa_airm_workspace_bytes(10, 0, 4, 7) == 0                          # bad dtype
    z = ctypes.c_void_p(0)
    args = lambda **kw: [kw.get("A", z), kw.get("nA", 4), z, 0, kw.get("m", 4), kw.get("dtype", 0),
                         1.0, 1e-6, 1, z, 0.0, kw.get("si", 0), kw.get("sc", 1), z, z, z, z, z, z,
                         kw.get("ws", z), 0, z]
    assert lib.sqfa_airm_pairwise(*args()) == -1                                     # null A
    fake = ctypes.c_void_p(4096)
    assert lib.sqfa_airm_pairwise(*args(A=fake)) == -1                               # null workspace
    assert lib.sqfa_airm_pairwise(*args(A=fake, ws=fake, dtype=5)) == -1             # dtype
    assert lib.sqfa_airm_pairwise(*args(A=fake, ws=fake, si=2, sc=2)) == -1          # shard
    assert lib.sqfa_airm_pairwise(*args(A=fake, ws=fake, m=1000)) == -2              # unsupported m
    assert lib.sqfa_airm_pairwise(*args(A=fake, ws=fake)) == -3                      # workspace too small
    assert b"workspace" in lib.sqfa_hip_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsqfa_hip.so")
    with pytest.raises(_lib.NativeLibraryError, match="no CPU fallback"):
        _lib.load()


def test_cpu_tensors_are_refused():
    import torch
    from sqfa_amd import distances
    S = torch.eye(3).repeat(4, 1, 1)
    with pytest.raises(RuntimeError, match="GPU only"):
        distances.affine_invariant(S, S)
