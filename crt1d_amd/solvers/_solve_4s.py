"""Tian et al. (2007) four-stream on MI355X; drop-in for ``crt1d/solvers/_solve_4s.py:8-293``."""
from .common import solve_single

short_name = "4s"
long_name = "Tian et al. four-stream"


def solve_4s(
    *, psi, I_dr0_all, I_df0_all, lai, leaf_t, leaf_r, soil_r, K_b_fn, G_fn, mu_s=0.501
):
    """The reference's two ``solve_bvp(tol=1e-6)`` runs per band are replaced by the exact solution of the
    same linear boundary-value problem, so results differ from the stock reference by its own BVP error
    (up to ~1e-4 relative) and agree with a tolerance-tightened reference to ~1e-10."""
    return solve_single("4s", psi=psi, I_dr0_all=I_dr0_all, I_df0_all=I_df0_all, lai=lai, leaf_t=leaf_t, leaf_r=leaf_r,
                        soil_r=soil_r, K_b_fn=K_b_fn, G_fn=G_fn, mu_s=mu_s)
