"""
ctypes binding of ``libcrt1d_hip.so`` (C ABI declared in ``include/crt1d_hip.h``).

There is deliberately NO fallback: if the HIP library is missing every solver raises.
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRT1D_HIP_LIB: measurement aid only -- an alternative build of the SAME library (tools/build_variant.sh) for in-round A/B runs
LIB_PATH = os.environ.get("CRT1D_HIP_LIB") or os.path.join(_HERE, "libcrt1d_hip.so")

# enum crt_scheme
SCHEME_IDS = {"2s": 0, "4s": 1, "n79": 2, "zq": 3, "bl": 4, "g77": 5, "bf": 6, "zq_pa": 7}
F32_SCHEMES = ("2s", "4s", "n79", "zq", "bl", "g77", "bf", "zq_pa")
TAU_D_METHODS = {"quad": 0, "9sky": 1}
NQ_TAU, NQ_G4, NQ_9SKY = 96, 32, 9
NQ = NQ_TAU + NQ_G4 + NQ_9SKY

FLAG_SKIP_PRECOMPUTE = 1
FLAG_PRECOMPUTE_ONLY = 2
FLAG_DIRECT_STORES = 4

CRT_OK = 0
CRT_ERR_BAD_ARG = -1
CRT_ERR_WORKSPACE = -2
CRT_ERR_UNSUPPORTED = -3
CRT_ERR_LAUNCH = -4
CRT_ERR_SHAPE = -5

_vp = ctypes.c_void_p


class CrtColumns(ctypes.Structure):
    _fields_ = [
        ("ncol", ctypes.c_int32),
        ("nz", ctypes.c_int32),
        ("psi", _vp),
        ("lai", _vp),
        ("mla", _vp),
        ("g_kind", _vp),
        ("g_param", _vp),
        ("g_at_psi", _vp),
        ("g_table", _vp),
    ]


class CrtBands(ctypes.Structure):
    _fields_ = [
        ("nb", ctypes.c_int32),
        ("col_stride", ctypes.c_int64),
        ("I_dr0", _vp),
        ("I_df0", _vp),
        ("leaf_r", _vp),
        ("leaf_t", _vp),
        ("soil_r", _vp),
    ]


NTUNE = 16


class CrtOptions(ctypes.Structure):
    _fields_ = [("mu_s", ctypes.c_double), ("tau_d_method", ctypes.c_int32), ("flags", ctypes.c_int32), ("tune", ctypes.c_int32 * NTUNE)]


class CrtOutputs(ctypes.Structure):
    _fields_ = [(k, _vp) for k in ("I_dr", "I_df_d", "I_df_u", "F", "x0", "x1", "x2")]


class CrtBandsumOut(ctypes.Structure):
    """``crt_bandsum_out``: layer-absorption band sums (+ ``totals``), and -- optional, all six or none -- the direct-beam part and the
    band-integrated level profiles of every irradiance variable ``diagnostics.band`` reduces."""

    _fields_ = [(k, _vp) for k in ("aI", "aI_sl", "aI_sh", "totals", "aI_dr", "I_dr", "I_df_d", "I_df_u", "F", "I_d")]


ABI_VERSION = 3


EXPORTS = [
    "crt_hip_abi_version",
    "crt_hip_strerror",
    "crt_hip_workspace_bytes",
    "crt_hip_workspace_bytes_nb",
    "crt_hip_quad_nodes",
    "crt_hip_solve_f64",
    "crt_hip_2s_f64",
    "crt_hip_4s_f64",
    "crt_hip_n79_f64",
    "crt_hip_zq_f64",
    "crt_hip_bl_f64",
    "crt_hip_g77_f64",
    "crt_hip_bf_f64",
    "crt_hip_zq_pa_f64",
    "crt_hip_solve_f32",
    "crt_hip_2s_f32",
    "crt_hip_4s_f32",
    "crt_hip_n79_f32",
    "crt_hip_zq_f32",
    "crt_hip_bl_f32",
    "crt_hip_g77_f32",
    "crt_hip_bf_f32",
    "crt_hip_zq_pa_f32",
    "crt_hip_absorb_bandsum_f64",
    "crt_hip_absorb_bandsum2_f64",
    "crt_hip_integrated_f64",
    "crt_hip_integrated2_f64",
    "crt_hip_absorb_f64",
    "crt_hip_band_reduce_f64",
    "crt_hip_tau_d_f64",
    "crt_hip_smear_tuv_f64",
    "crt_hip_lai_beta_f64",
    "crt_hip_buffer_alloc_set",
    "crt_hip_buffer_alloc",
    "crt_hip_buffer_free",
    "crt_hip_buffer_trim",
    "crt_hip_buffer_set_retain",
    "crt_hip_buffer_describe",
    "crt_hip_buffer_stats",
    "crt_hip_last_kernel",
    "crt_hip_probe_fill_f64",
    "crt_hip_probe_copy_f64",
    "crt_hip_probe_store_set_f64",
    "crt_hip_probe_math_f64",
]

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises :class:`HipLibraryMissing` if not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: the HIP kernels are not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C crt1d_amd/csrc`). "
            "crt1d_amd has no CPU fallback."
        )
    # torch first: it brings its own libamdhip64.so.7; ours must bind to the same runtime instance
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    lib.crt_hip_abi_version.restype = ctypes.c_int
    lib.crt_hip_strerror.restype = ctypes.c_char_p
    lib.crt_hip_strerror.argtypes = [ctypes.c_int]
    lib.crt_hip_workspace_bytes.restype = ctypes.c_size_t
    lib.crt_hip_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int32, ctypes.c_int32]
    lib.crt_hip_workspace_bytes_nb.restype = ctypes.c_size_t
    lib.crt_hip_workspace_bytes_nb.argtypes = [ctypes.c_int, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    lib.crt_hip_quad_nodes.restype = ctypes.c_int
    lib.crt_hip_quad_nodes.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    solve_args = [
        ctypes.POINTER(CrtColumns),
        ctypes.POINTER(CrtBands),
        ctypes.POINTER(CrtOptions),
        ctypes.POINTER(CrtOutputs),
        _vp,
        ctypes.c_size_t,
        _vp,
    ]
    lib.crt_hip_solve_f64.restype = ctypes.c_int
    lib.crt_hip_solve_f64.argtypes = [ctypes.c_int] + solve_args
    lib.crt_hip_solve_f32.restype = ctypes.c_int
    lib.crt_hip_solve_f32.argtypes = [ctypes.c_int] + solve_args
    for s in SCHEME_IDS:
        for suffix in ("f64", "f32"):  # the f32 structs share the f64 layout (include/crt1d_hip.h)
            if suffix == "f32" and s not in F32_SCHEMES:
                continue
            f = getattr(lib, f"crt_hip_{s}_{suffix}")
            f.restype = ctypes.c_int
            f.argtypes = solve_args
    lib.crt_hip_absorb_bandsum_f64.restype = ctypes.c_int
    lib.crt_hip_absorb_bandsum_f64.argtypes = [
        ctypes.POINTER(CrtColumns), ctypes.POINTER(CrtBands), _vp, _vp, _vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp,
    ]
    lib.crt_hip_integrated_f64.restype = ctypes.c_int
    lib.crt_hip_integrated_f64.argtypes = [
        ctypes.c_int, ctypes.POINTER(CrtColumns), ctypes.POINTER(CrtBands), ctypes.POINTER(CrtOptions), _vp, ctypes.c_int32,
        _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp,
    ]
    lib.crt_hip_absorb_bandsum2_f64.restype = ctypes.c_int
    lib.crt_hip_absorb_bandsum2_f64.argtypes = [
        ctypes.POINTER(CrtColumns), ctypes.POINTER(CrtBands), _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.POINTER(CrtBandsumOut), _vp,
    ]
    lib.crt_hip_integrated2_f64.restype = ctypes.c_int
    lib.crt_hip_integrated2_f64.argtypes = [
        ctypes.c_int, ctypes.POINTER(CrtColumns), ctypes.POINTER(CrtBands), ctypes.POINTER(CrtOptions), _vp, ctypes.c_int32,
        ctypes.POINTER(CrtBandsumOut), _vp, ctypes.c_size_t, _vp,
    ]
    lib.crt_hip_band_reduce_f64.restype = ctypes.c_int
    lib.crt_hip_band_reduce_f64.argtypes = [_vp, ctypes.c_int64, ctypes.c_int32, _vp, ctypes.c_int32, _vp, _vp]
    lib.crt_hip_absorb_f64.restype = ctypes.c_int
    lib.crt_hip_absorb_f64.argtypes = [ctypes.POINTER(CrtColumns), ctypes.POINTER(CrtBands), _vp, _vp, _vp, ctypes.POINTER(_vp), _vp, _vp, _vp]
    lib.crt_hip_tau_d_f64.restype = ctypes.c_int
    lib.crt_hip_tau_d_f64.argtypes = [_vp, _vp, ctypes.c_int64, ctypes.c_int32, _vp, _vp]
    lib.crt_hip_smear_tuv_f64.restype = ctypes.c_int
    lib.crt_hip_smear_tuv_f64.argtypes = [_vp, ctypes.c_int64, ctypes.c_int32, _vp, ctypes.c_int32, _vp, ctypes.c_int32, _vp, _vp]
    lib.crt_hip_lai_beta_f64.restype = ctypes.c_int
    lib.crt_hip_lai_beta_f64.argtypes = [_vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp, _vp]
    lib.crt_hip_buffer_alloc.restype = ctypes.c_int
    lib.crt_hip_buffer_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    lib.crt_hip_buffer_free.restype = ctypes.c_int
    lib.crt_hip_buffer_free.argtypes = [ctypes.c_void_p]
    lib.crt_hip_buffer_alloc_set.restype = ctypes.c_int
    lib.crt_hip_buffer_alloc_set.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_void_p)]
    lib.crt_hip_buffer_trim.restype = ctypes.c_int
    lib.crt_hip_buffer_trim.argtypes = []
    lib.crt_hip_buffer_set_retain.restype = ctypes.c_int
    lib.crt_hip_buffer_set_retain.argtypes = [ctypes.c_size_t]
    lib.crt_hip_buffer_describe.restype = ctypes.c_int
    lib.crt_hip_buffer_describe.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    lib.crt_hip_buffer_stats.restype = ctypes.c_int
    lib.crt_hip_buffer_stats.argtypes = [ctypes.POINTER(ctypes.c_int64)]
    lib.crt_hip_last_kernel.restype = ctypes.c_char_p
    lib.crt_hip_last_kernel.argtypes = []
    lib.crt_hip_probe_fill_f64.restype = ctypes.c_int
    lib.crt_hip_probe_fill_f64.argtypes = [_vp, ctypes.c_size_t, ctypes.c_double, _vp]
    lib.crt_hip_probe_copy_f64.restype = ctypes.c_int
    lib.crt_hip_probe_copy_f64.argtypes = [_vp, _vp, ctypes.c_size_t, _vp]
    lib.crt_hip_probe_math_f64.restype = ctypes.c_int
    lib.crt_hip_probe_math_f64.argtypes = [_vp, ctypes.c_size_t, _vp, _vp, _vp, _vp]
    lib.crt_hip_probe_store_set_f64.restype = ctypes.c_int
    lib.crt_hip_probe_store_set_f64.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_double, _vp]
    if lib.crt_hip_abi_version() != ABI_VERSION:
        raise HipLibraryMissing(f"{LIB_PATH}: ABI version mismatch, rebuild")
    _lib = lib
    return lib


def strerror(status):
    return load().crt_hip_strerror(int(status)).decode()


def check(status, what):
    """Translate a C-ABI status into the reference's exception conventions
    (AssertionError for shape/orientation, ValueError for bad options)."""
    if status == CRT_OK:
        return
    msg = f"{what}: {strerror(status)} (status {status})"
    if status == CRT_ERR_SHAPE:
        raise AssertionError(msg)
    if status == CRT_ERR_BAD_ARG:
        raise ValueError(msg)
    raise RuntimeError(msg)


def quad_nodes(mu_s=0.501):
    """Zenith angles (radians) at which a sampled ``G_fn`` table must be given; (NQ,) float64."""
    import numpy as np

    buf = (ctypes.c_double * NQ)()
    check(load().crt_hip_quad_nodes(float(mu_s), buf), "crt_hip_quad_nodes")
    return np.frombuffer(buf, dtype=np.float64).copy()
