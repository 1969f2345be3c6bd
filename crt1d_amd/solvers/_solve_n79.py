"""Norman (1979) on MI355X; drop-in for ``crt1d/solvers/_solve_n79.py:11-164``."""
from .common import solve_single

short_name = "N79"
long_name = "Norman (1979)"


def solve_n79(
    *,
    psi,
    I_dr0_all, I_df0_all,
    lai,
    leaf_t, leaf_r,
    soil_r,
    K_b_fn,
    tau_d_method="quad",  # set to '9sky' to compare to Bonan
):
    """Returns ``I_dr, I_df_d, I_df_u, F`` ``(n_z, n_wl)`` and ``aI_lsl, aI_lsh`` ``(n_z-1, n_wl)``.
    Raises ``ValueError`` for an unknown ``tau_d_method`` (``common.py:78``)."""
    return solve_single("n79", psi=psi, I_dr0_all=I_dr0_all, I_df0_all=I_df0_all, lai=lai, leaf_t=leaf_t, leaf_r=leaf_r,
                        soil_r=soil_r, K_b_fn=K_b_fn, tau_d_method=tau_d_method)
