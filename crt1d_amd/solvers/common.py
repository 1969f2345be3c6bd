"""
Host side of the single-column plugin functions: turn the reference's keyword arguments
(NumPy arrays + Python callables) into device buffers, call the batched HIP path with ``ncol = 1`` and hand
NumPy ``(nz, nb)`` arrays back -- the ownership/return conventions of ``crt1d/solvers/_solve_*.py``.

``K_b_fn`` / ``G_fn`` are arbitrary callables in the reference (``variables.yml:137-162``).  A kernel cannot call
them, so: a :class:`crt1d_amd.leaf_angle.GFunction` (or anything tagged with one) is evaluated in closed form on
device; any other callable is sampled on the host at the library's fixed quadrature nodes
(``crt_hip_quad_nodes``) and shipped as a table.
"""

import math
import warnings

import numpy as np

from .. import _lib, batched
from ..leaf_angle import G_TABLE, describe_G


class KbFunction:
    """``K_b_fn(psi) = G_fn(psi) / cos(psi)`` (``crt1d/model.py:291``) that remembers its ``G_fn``."""

    def __init__(self, G_fn):
        self.G_fn = G_fn
        d = describe_G(G_fn)
        if d is not None:
            self.gfunction = getattr(G_fn, "gfunction", G_fn)

    def __call__(self, psi):
        return self.G_fn(psi) / np.cos(psi)


def tau_b_fn(K_b_fn, psi, lai):
    """Direct-beam transmittance ``exp(-K_b(psi) lai)`` (``crt1d/solvers/common.py:11-27``); a one-line closed form, evaluated
    where the arguments live."""
    return np.exp(-K_b_fn(psi) * lai)


def tau_df_fn(K_b_fn, lai, *, method="quad"):
    """Diffuse transmittance of foliage with LAI ``lai`` (scalar or array), drop-in for ``crt1d/solvers/common.py:56-87``.
    ``K_b_fn`` is sampled at the library's quadrature angles and the integral is formed on the device
    (``crt_hip_tau_d_f64``: fixed 96-node rule for 'quad', the reference's nine angles for '9sky')."""
    import torch

    if method not in _lib.TAU_D_METHODS:
        raise ValueError("invalid `method`. Valid options are 'quad' and '9sky'.")  # common.py:78
    lib = _lib.load()
    nodes = _lib.quad_nodes()
    kb = np.ascontiguousarray(_sample(K_b_fn, nodes))
    L = np.atleast_1d(np.asarray(lai, dtype=np.float64))
    dev = torch.device("cuda", torch.cuda.current_device())
    kb_d, L_d = torch.as_tensor(kb).to(dev), torch.as_tensor(np.ascontiguousarray(L.reshape(-1))).to(dev)
    out = torch.empty_like(L_d)
    with torch.cuda.device(dev):
        st = lib.crt_hip_tau_d_f64(kb_d.data_ptr(), L_d.data_ptr(), L_d.numel(), _lib.TAU_D_METHODS[method], out.data_ptr(),
                                   torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(st, "crt_hip_tau_d_f64")
    res = out.cpu().numpy().reshape(L.shape)
    return float(res[0]) if np.isscalar(lai) else res


def K_df_fn(K_b_fn, lai_tot, **kwargs):
    """``K_d = -ln(tau_d(LAI)) / LAI`` (``crt1d/solvers/common.py:90-95``)."""
    return -np.log(tau_df_fn(K_b_fn, lai_tot, **kwargs)) / lai_tot


def _sample(fn, psi):
    """Evaluate a user callable on an array of angles; falls back to a scalar loop."""
    try:
        r = np.asarray(fn(psi), dtype=np.float64)
        if r.shape == psi.shape:
            return r
        if r.ndim == 0:  # constant-returning callables such as G_spherical
            probe = [float(fn(float(p))) for p in (psi[0], psi[-1])]
            if probe[0] == probe[1] == float(r):
                return np.full(psi.shape, float(r))
    except Exception:
        pass
    return np.array([float(fn(float(p))) for p in psi], dtype=np.float64)


def _describe(psi, K_b_fn, G_fn, mu_s):
    """-> dict(g_kind, g_param, g_at_psi, g_table) for one column."""
    d = None
    if G_fn is not None:
        d = describe_G(G_fn)
    if d is None and K_b_fn is not None:
        d = describe_G(getattr(K_b_fn, "gfunction", None)) if hasattr(K_b_fn, "gfunction") else None
    if d is not None:
        return dict(g_kind=d[0], g_param=d[1], g_at_psi=None, g_table=None)
    nodes = _lib.quad_nodes(mu_s)
    if G_fn is not None:
        table = _sample(G_fn, nodes)
    else:
        table = _sample(K_b_fn, nodes) * np.cos(nodes)
    K_b = float(K_b_fn(psi)) if K_b_fn is not None else float(G_fn(psi)) / math.cos(psi)
    g_at_psi = K_b * math.cos(psi)
    if G_fn is not None and K_b_fn is not None:
        g2 = float(G_fn(psi))
        if abs(g2 - g_at_psi) > 1e-12 * max(1.0, abs(g2)):
            warnings.warn("`K_b_fn(psi) * cos(psi)` and `G_fn(psi)` disagree; the device path uses `K_b_fn(psi)`.")
    return dict(g_kind=G_TABLE, g_param=0.0, g_at_psi=g_at_psi, g_table=table)


def solve_single(scheme, *, psi, I_dr0_all, I_df0_all, lai, leaf_t, leaf_r, soil_r=None, K_b_fn=None, G_fn=None, mla=None,
                 mu_s=0.501, tau_d_method="quad"):
    import torch

    if tau_d_method not in _lib.TAU_D_METHODS:
        raise ValueError("invalid `method`. Valid options are 'quad' and '9sky'.")  # common.py:78
    _lib.load()  # raises if the HIP library is not built
    if not torch.cuda.is_available():
        raise RuntimeError("crt1d_amd runs on an AMD GPU only (no CPU fallback); torch.cuda.is_available() is False")
    lai = np.ascontiguousarray(lai, dtype=np.float64)
    if lai.ndim != 1:
        raise ValueError("`lai` must be 1-D (n_z,)")
    spectra = {}
    for name, v in (("I_dr0", I_dr0_all), ("I_df0", I_df0_all), ("leaf_r", leaf_r), ("leaf_t", leaf_t), ("soil_r", soil_r)):
        if v is None:
            continue
        v = np.ascontiguousarray(v, dtype=np.float64)
        if v.ndim != 1:
            raise ValueError(f"`{name}` must be 1-D (n_wl,)")
        spectra[name] = v
    nb = spectra["I_dr0"].size
    if any(v.size != nb for v in spectra.values()):
        raise ValueError("spectral inputs must all have size n_wl")
    psi = float(psi)
    g = _describe(psi, K_b_fn, G_fn, mu_s)
    dev = torch.device("cuda", torch.cuda.current_device())
    t = lambda a: torch.as_tensor(np.atleast_1d(np.asarray(a, dtype=np.float64))).to(dev)  # noqa: E731
    cols = batched.Columns(
        psi=t(psi), lai=t(lai)[None, :], g_kind=torch.tensor([g["g_kind"]], dtype=torch.int32, device=dev), g_param=t(g["g_param"]),
        mla=None if mla is None else t(float(mla)),
        g_at_psi=None if g["g_at_psi"] is None else t(g["g_at_psi"]),
        g_table=None if g["g_table"] is None else t(g["g_table"])[None, :],
    )
    bands = batched.Bands(t(spectra["I_dr0"]), t(spectra["I_df0"]), t(spectra["leaf_r"]), t(spectra["leaf_t"]),
                          None if "soil_r" not in spectra else t(spectra["soil_r"]))
    sol = batched.solve(scheme, cols, bands, mu_s=mu_s, tau_d_method=tau_d_method)
    return {k: v[0].cpu().numpy() for k, v in sol.items()}
