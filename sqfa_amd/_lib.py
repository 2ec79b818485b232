"""ctypes loader for libsqfa_hip.so (the C ABI declared in include/sqfa_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C sqfa_amd/csrc``.
There is no fallback: if it cannot be loaded, every native entry point raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SQFA_HIP_LIBRARY selects another build of the same library (development: A/B timing of kernel variants,
# tools/build_variant.sh) without touching the installed file.
LIB_PATH = os.environ.get("SQFA_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libsqfa_hip.so")

SQFA_F32, SQFA_F64 = 0, 1
SQFA_OK = 0
STATUS_TEXT = {
    -1: "bad argument",
    -2: "matrix size not supported by the native kernels",
    -3: "workspace too small",
    -4: "HIP launch failed",
}

class AirmOptions(ctypes.Structure):
    """sqfa_airm_options (include/sqfa_hip.h): per-call policies of sqfa_airm_pairwise_opt."""
    _fields_ = [("geometry_policy", ctypes.c_int), ("class_factor_policy", ctypes.c_int),
                ("sweep_counter", ctypes.c_void_p), ("mean_metric_policy", ctypes.c_int)]


_PAIRWISE_ARGS = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                  ctypes.c_double, ctypes.c_double, ctypes.c_int,
                  ctypes.c_void_p, ctypes.c_double,
                  ctypes.c_int, ctypes.c_int,
                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                  ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]

# every symbol include/sqfa_hip.h declares: name -> (restype, argtypes)
_c_int_p = ctypes.POINTER(ctypes.c_int)
PROTOTYPES = {
    "sqfa_hip_version": (ctypes.c_int, []),
    "sqfa_hip_arch": (ctypes.c_char_p, []),
    "sqfa_hip_max_dim": (ctypes.c_int, []),
    "sqfa_hip_last_error": (ctypes.c_char_p, []),
    "sqfa_airm_tiling": (ctypes.c_int, [ctypes.c_int] * 4 + [_c_int_p] * 5),
    "sqfa_airm_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "sqfa_airm_workspace_bytes_sharded": (ctypes.c_size_t, [ctypes.c_int] * 6),
    "sqfa_airm_pairwise": (ctypes.c_int, list(_PAIRWISE_ARGS)),
    "sqfa_airm_pairwise_opt": (ctypes.c_int, list(_PAIRWISE_ARGS) + [ctypes.POINTER(AirmOptions)]),
    "sqfa_airm_eigenvalues_backward": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
         ctypes.POINTER(AirmOptions)],
    ),
    "sqfa_gauss_pair_terms": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
         ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "sqfa_project_scatters": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_void_p],
    ),
    "sqfa_packed_scatter_elems": (ctypes.c_size_t, [ctypes.c_int]),
    "sqfa_pack_scatters": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "sqfa_project_scatters_packed": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_void_p],
    ),
    "sqfa_feature_scatters": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_void_p],
    ),
    "sqfa_feature_scatters_backward": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_void_p],
    ),
    "sqfa_feature_scatters_ex": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "sqfa_feature_scatters_backward_ex": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
         ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "sqfa_embed_backward_means": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "sqfa_sphere_forward": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "sqfa_sphere_backward": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "sqfa_lbfgs_max_history": (ctypes.c_int, []),
    "sqfa_lbfgs_work_elems": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "sqfa_lbfgs_push": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p],
    ),
    "sqfa_lbfgs_step_stats": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p],
    ),
    "sqfa_lbfgs_direction": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
         ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p],
    ),
    "sqfa_spd_function_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "sqfa_spd_function": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p],
    ),
    "sqfa_spd_function_backward": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
         ctypes.c_void_p, ctypes.c_void_p],
    ),
    "sqfa_airm_profile": (ctypes.c_int, [ctypes.c_int]),
    "sqfa_airm_profile_read": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), _c_int_p]),
    "sqfa_project_profile_read": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), _c_int_p]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the shared library, with prototypes attached."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C sqfa_amd/csrc`. sqfa_amd has no CPU fallback for the pairwise distance path."
        )
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as err:  # e.g. libamdhip64 not found
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {err}") from err
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status, what):
    if status != SQFA_OK:
        detail = load().sqfa_hip_last_error().decode() if _lib is not None else ""
        raise NativeLibraryError(f"{what} failed: {STATUS_TEXT.get(status, status)} ({detail})")
