"""
Seeded synthetic canopy columns x bands (SURVEY.md section 8(d) generator).

Value ranges bracket the reference's default case (``crt1d/cases.py:15-58``):
equal-dLAI cumulative profiles like every reference generator
(``crt1d/leaf_area.py:82``), ellipsoidal-approx leaf angles, per-(column, band)
leaf/soil optics and top-of-canopy irradiances.

Pure NumPy, host side; ``bench.py``/tests move the arrays to the GPU.
"""

import numpy as np

from .leaf_angle import G_ELLIPSOIDAL_APPROX, mla_to_x_approx


def make_columns(ncol, nb, nz, *, seed=1234, uniform_dlai=True, per_column_optics=True, dtype=np.float64):
    """Return a dict of host arrays describing ``ncol`` columns x ``nb`` bands.

    Keys: ``psi (ncol,)``, ``lai (ncol,nz)`` (index 0 = ground = total LAI, -1 = top = 0),
    ``mla (ncol,)``, ``g_kind (ncol,) int32``, ``g_param (ncol,)``,
    ``leaf_r leaf_t soil_r I_dr0 I_df0`` each ``(ncol,nb)`` (or ``(1,nb)`` when
    ``per_column_optics`` is False -> broadcast over columns), ``wle (nb+1,)`` band edges (um).
    """
    rng = np.random.default_rng(seed)
    lai_tot = rng.uniform(0.5, 8.0, ncol)
    if uniform_dlai:
        frac = np.linspace(1.0, 0.0, nz)[None, :]
        frac = np.broadcast_to(frac, (ncol, nz))
    else:
        gamma = rng.uniform(0.5, 2.0, ncol)
        frac = 1.0 - np.linspace(0.0, 1.0, nz)[None, :] ** gamma[:, None]
        frac[:, 0] = 1.0
        frac[:, -1] = 0.0
    lai = lai_tot[:, None] * frac
    psi = np.deg2rad(rng.uniform(0.0, 75.0, ncol))
    mla = rng.uniform(20.0, 80.0, ncol)
    x = mla_to_x_approx(mla)

    nc_opt = ncol if per_column_optics else 1
    leaf_r = rng.uniform(0.02, 0.55, (nc_opt, nb))
    leaf_t = rng.uniform(0.02, 0.45, (nc_opt, nb))
    s = leaf_r + leaf_t
    scale = np.where(s > 0.95, 0.95 / s, 1.0)
    leaf_r = leaf_r * scale
    leaf_t = leaf_t * scale
    soil_r = rng.uniform(0.05, 0.40, (nc_opt, nb))
    I_dr0 = rng.uniform(0.0, 10.0, (nc_opt, nb))
    I_df0 = rng.uniform(0.0, 5.0, (nc_opt, nb))
    wle = np.linspace(0.3, 2.6, nb + 1)

    f = lambda a: np.ascontiguousarray(a, dtype=dtype)  # noqa: E731
    return dict(
        psi=f(psi),
        lai=f(lai),
        mla=f(mla),
        g_kind=np.full(ncol, G_ELLIPSOIDAL_APPROX, dtype=np.int32),
        g_param=f(x),
        leaf_r=f(leaf_r),
        leaf_t=f(leaf_t),
        soil_r=f(soil_r),
        I_dr0=f(I_dr0),
        I_df0=f(I_df0),
        wle=wle,
    )
