"""Dickinson-Sellers two-stream on MI355X; drop-in for ``crt1d/solvers/_solve_2s.py:11-163``."""
from .common import solve_single

short_name = "2s"
long_name = "Dickinson–Sellers two-stream"


def solve_2s(
    *,
    psi,
    I_dr0_all, I_df0_all,
    lai,
    leaf_t, leaf_r,
    soil_r,
    K_b_fn, G_fn, mla,
):
    """Same keyword-only signature, returns ``dict(I_dr, I_df_d, I_df_u, F)`` of fresh ``(n_z, n_wl)`` arrays.
    ``mu_bar`` (``_solve_2s.py:32``) is integrated with fixed Gauss-Legendre nodes instead of QUADPACK."""
    return solve_single("2s", psi=psi, I_dr0_all=I_dr0_all, I_df0_all=I_df0_all, lai=lai, leaf_t=leaf_t, leaf_r=leaf_r,
                        soil_r=soil_r, K_b_fn=K_b_fn, G_fn=G_fn, mla=mla)
