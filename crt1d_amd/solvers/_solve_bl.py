"""Beer-Lambert on MI355X; drop-in for ``crt1d/solvers/_solve_bl.py:9-93``."""
from .common import solve_single

short_name = "B–L"
long_name = "Beer–Lambert"


def solve_bl(
    *,
    psi,
    I_dr0_all,
    I_df0_all,
    lai,
    leaf_t,
    leaf_r,
    K_b_fn,
):
    return solve_single("bl", psi=psi, I_dr0_all=I_dr0_all, I_df0_all=I_df0_all, lai=lai, leaf_t=leaf_t, leaf_r=leaf_r,
                        K_b_fn=K_b_fn)
