"""Zhao & Qualls multi-scattering, pyAPES variant, on MI355X; drop-in for ``crt1d/solvers/_solve_zq_pa.py:24-418``."""
from .common import solve_single

short_name = "ZQ-pA"
long_name = "Zhao & Qualls multi-scattering (pyAPES)"


def solve_zq_pa(
    *,
    psi,
    I_dr0_all,
    I_df0_all,
    lai,
    clump,
    leaf_t,
    leaf_r,
    soil_r,
    K_b_fn,
):
    """The zq system on M = min(100, n_z) equal layers, interpolated back to ``lai`` (n_wl <= 1024 for now).
    ``clump`` only enters absorption terms the reference computes but does not return (``_solve_zq_pa.py:364-404``)."""
    return solve_single("zq_pa", psi=psi, I_dr0_all=I_dr0_all, I_df0_all=I_df0_all, lai=lai, leaf_t=leaf_t, leaf_r=leaf_r,
                        soil_r=soil_r, K_b_fn=K_b_fn)
