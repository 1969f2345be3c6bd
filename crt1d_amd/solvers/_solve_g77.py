"""Goudriaan (1977) on MI355X; drop-in for ``crt1d/solvers/_solve_g77.py:7-135``."""
import numpy as np

from .common import solve_single

short_name = "G77"
long_name = "Goudriaan (1977)"


def solve_g77(
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
    lai = np.asarray(lai)
    assert lai[0] == lai.max()  # _solve_g77.py:35
    return solve_single("g77", psi=psi, I_dr0_all=I_dr0_all, I_df0_all=I_df0_all, lai=lai, leaf_t=leaf_t, leaf_r=leaf_r,
                        soil_r=soil_r, K_b_fn=K_b_fn)
