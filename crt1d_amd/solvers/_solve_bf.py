"""Bodin & Franklin improved Goudriaan on MI355X; drop-in for ``crt1d/solvers/_solve_bf.py:7-154``."""
import numpy as np

from .common import solve_single

short_name = "BF"
long_name = "Bodin & Franklin improved Goudriaan"


def solve_bf(
    *,
    psi,
    I_dr0_all,
    I_df0_all,
    lai,
    leaf_t,
    leaf_r,
    soil_r,
    K_b_fn,
):
    """As the reference, the extra ``rho_c`` entry is the canopy reflectance of the LAST band only
    (``_solve_bf.py:78,153``); it is a closed form of that band's optics and is evaluated on the host."""
    lai = np.asarray(lai)
    assert lai[0] == lai.max()  # _solve_bf.py:40
    sol = solve_single("bf", psi=psi, I_dr0_all=I_dr0_all, I_df0_all=I_df0_all, lai=lai, leaf_t=leaf_t, leaf_r=leaf_r,
                       soil_r=soil_r, K_b_fn=K_b_fn)
    k_prime = np.sqrt(1 - (np.asarray(leaf_r)[-1] + np.asarray(leaf_t)[-1]))
    sol["rho_c"] = ((1 - k_prime) / (1 + k_prime)) * (2 / (1 + 1.6 * np.cos(psi)))
    return sol
