"""
(Complete) sets of input parameters -- the default canopy of the reference (``crt1d/cases.py:15-58``):
h_c = 20 m, LAI = 4, mean leaf angle 57 deg (ellipsoidal approx.), SZA = 20 deg, equal-dLAI beta leaf-area profile
(``crt1d/leaf_area.py:42-93``) and the default spectra (ideal green leaf, SPCTRAL2 default spectrum binned to in-band
W m-2, two-value soil; ``crt1d/data/__init__.py:23-35,90-157,188-217``) with NaN bands dropped -> 107 bands.

The spectra ship as ``crt1d_amd/data/default_spectra.npz`` (arrays only, produced by ``oracle/gen_golden.py``).
"""

import os

import numpy as np

from .leaf_angle import G_ELLIPSOIDAL_APPROX, GFunction, mla_to_x_approx

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "default_spectra.npz")


def distribute_lai_beta(h_c, LAI, n, *, h_min=0.5):
    """Equal-dLAI cumulative profile with heights from a beta leaf-area-density distribution whose mode sits at
    0.7 h_c (``crt1d/leaf_area.py:42-93``).  Returns ``(lai, z)``; ``lai`` decreases with ``z`` (index 0 = ground)."""
    from scipy.stats import beta

    d = (h_c - 0.7 * h_c) / h_c  # relative depth of maximum LAD = desired mode
    b = 3
    a = -((b - 2) * d + 1) / (d - 1)
    frac = np.linspace(1.0, 0, n)
    z = (h_c - h_min) * (1 - beta(a, b).ppf(frac)) + h_min
    return frac * LAI, z


def load_default_case(nlayers):
    """Idealized beta leaf distribution + default spectra; same keys as the reference's dict."""
    lai, z = distribute_lai_beta(20.0, 4.0, nlayers)
    sp = np.load(_DATA)
    mla = 57  # approximately the value for the spherical leaf angle distribution
    orient = float(mla_to_x_approx(mla))
    G_fn = GFunction(G_ELLIPSOIDAL_APPROX, orient)
    return dict(
        lai=lai, z=z, green=1.0, mla=mla, clump=1.0, orient=orient, G_fn=G_fn, psi=np.deg2rad(20),
        leaf_t=sp["leaf_t"].copy(), leaf_r=sp["leaf_r"].copy(), soil_r=sp["soil_r"].copy(), wl_leafsoil=sp["wl"].copy(),
        I_dr0_all=sp["I_dr0_all"].copy(), I_df0_all=sp["I_df0_all"].copy(), wl=sp["wl"].copy(), dwl=sp["dwl"].copy(),
    )
