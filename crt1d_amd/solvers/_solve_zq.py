"""Zhao & Qualls (2005) multi-scattering on MI355X; drop-in for ``crt1d/solvers/_solve_zq.py:13-229``."""
from .common import solve_single

short_name = "ZQ"
long_name = "Zhao & Qualls multi-scattering"


def solve_zq(
    *, psi,
    I_dr0_all, I_df0_all,
    lai,
    leaf_t, leaf_r, soil_r,
    K_b_fn, G_fn,
):
    """Returns the four standard profiles plus the single-scattering ones ``I_df_d_ss, I_df_u_ss, F_ss``."""
    return solve_single("zq", psi=psi, I_dr0_all=I_dr0_all, I_df0_all=I_df0_all, lai=lai, leaf_t=leaf_t, leaf_r=leaf_r,
                        soil_r=soil_r, K_b_fn=K_b_fn, G_fn=G_fn)
