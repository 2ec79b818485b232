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


def test_argument_validation_of_the_round2_entry_points():
    """The round-2 entry points reject bad arguments on the host, before any launch (no GPU needed)."""
    lib = _lib.load()
    z, fake = ctypes.c_void_p(0), ctypes.c_void_p(4096)
    # eigenvalue backward: needs weights and an output
    assert lib.sqfa_airm_eigenvalues_backward(fake, 4, fake, 4, 4, 0, z, fake, fake, fake, 1 << 30, z, None) == -1
    assert lib.sqfa_airm_eigenvalues_backward(fake, 4, fake, 4, 4, 0, fake, z, fake, fake, 1 << 30, z, None) == -1
    # Gaussian pair terms: null inputs, bad dtype, too large, gradient outputs without upstream gradients
    assert lib.sqfa_gauss_pair_terms(z, fake, 3, fake, fake, 3, 4, 0, z, z, fake, fake, z, z, z) == -1
    assert lib.sqfa_gauss_pair_terms(fake, fake, 3, fake, fake, 3, 4, 9, z, z, fake, fake, z, z, z) == -1
    assert lib.sqfa_gauss_pair_terms(fake, fake, 3, fake, fake, 3, 65, 0, z, z, fake, fake, z, z, z) == -2
    assert lib.sqfa_gauss_pair_terms(fake, fake, 3, fake, fake, 3, 4, 0, z, z, z, z, fake, fake, z) == -1
    assert lib.sqfa_gauss_pair_terms(fake, fake, 3, fake, fake, 3, 4, 0, fake, z, z, z, fake, z, z) == -1
    # closure glue
    assert lib.sqfa_sphere_forward(z, 4, 8, 0, fake, fake, z) == -1
    assert lib.sqfa_sphere_forward(fake, 4, 8, 5, fake, fake, z) == -1
    assert lib.sqfa_sphere_backward(fake, fake, 4, 8, 0, z, 3, z, z, fake, z) == -1          # groups without partial sums
    assert lib.sqfa_embed_backward_means(fake, z, 3, 4, 0, fake, z) == -1
    assert lib.sqfa_feature_scatters_ex(fake, 4, 10, fake, 3, 0, 0.0, z, fake, z) == -2       # D % 4 != 0
    assert lib.sqfa_feature_scatters_backward_ex(fake, 3, fake, 3, 8, 4, 0, 2, 0, fake, z) == -1  # ldg < K
    assert lib.sqfa_feature_scatters_backward_ex(fake, 70, fake, 3, 8, 65, 0, 2, 1, fake, z) == -2  # K > 64
    # L-BFGS
    assert lib.sqfa_lbfgs_max_history() >= 100
    assert lib.sqfa_lbfgs_step_stats(fake, fake, fake, 1.0, 0, fake, fake, fake, fake, 0, z) == -1   # n < 1
    assert lib.sqfa_lbfgs_work_elems(100, 50000) >= 3 * 100 + 50000 and lib.sqfa_lbfgs_work_elems(500, 10) == 0
    assert lib.sqfa_lbfgs_push(fake, fake, fake, 200, 10, 0, fake, fake, fake, 0, z) == -1          # history too long
    assert lib.sqfa_lbfgs_push(fake, fake, fake, 10, 10, 10, fake, fake, fake, 0, z) == -1          # slot out of range
    slots = (ctypes.c_int * 2)(0, 11)
    assert lib.sqfa_lbfgs_direction(fake, fake, fake, 10, 10, slots, 2, fake, z, fake, fake, 0, z) == -1   # bad slot
    # per-shard workspace: never more than the any-shard bound, and decreasing with the shard's share
    any_shards = lib.sqfa_airm_workspace_bytes(1000, 0, 16, 0)
    one, eight = lib.sqfa_airm_workspace_bytes_sharded(1000, 0, 16, 0, 1, 0), lib.sqfa_airm_workspace_bytes_sharded(1000, 0, 16, 0, 8, 0)
    assert 0 < eight < one <= any_shards and one < 60e6
    assert lib.sqfa_airm_workspace_bytes_sharded(1000, 0, 16, 0, 0, 0) == 0
    # the any-policy size covers either lane geometry of a small launch, whichever policy a call will carry
    for m in (8, 12, 16, 17, 20, 32):
        for dt in (0, 1):
            anyp = lib.sqfa_airm_workspace_bytes(30, 0, m, dt)
            assert anyp >= max(lib.sqfa_airm_workspace_bytes_sharded(30, 0, m, dt, 1, pol) for pol in (-1, 0, 1)) > 0
    # per-class matrix functions (round 4)
    assert lib.sqfa_spd_function_workspace_bytes(10, 16, 0) >= 10 * 16 * 16 * 4
    assert lib.sqfa_spd_function_workspace_bytes(10, 65, 0) == 0 and lib.sqfa_spd_function_workspace_bytes(0, 4, 0) == 0
    assert lib.sqfa_spd_function(z, 3, 4, 0, 0, fake, fake, fake, fake, 1 << 20, z) == -1            # null S
    assert lib.sqfa_spd_function(fake, 3, 4, 0, 7, fake, fake, fake, fake, 1 << 20, z) == -1         # kind
    assert lib.sqfa_spd_function(fake, 3, 65, 0, 0, fake, fake, fake, fake, 1 << 20, z) == -2        # m > 64
    assert lib.sqfa_spd_function(fake, 3, 4, 0, 0, fake, fake, fake, fake, 8, z) == -3               # workspace
    assert lib.sqfa_spd_function_backward(fake, fake, z, 3, 4, 0, 0, fake, z) == -1                  # null G
    assert lib.sqfa_spd_function_backward(fake, fake, fake, 3, 65, 0, 0, fake, z) == -2
