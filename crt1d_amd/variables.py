"""
Argument contract of the solver plugins.

The reference derives these lists from ``crt1d/variables.yml`` (``intent: in`` entries at
:37-170, ``intent: out`` at :175-208) through ``Vmd.intent`` (``crt1d/variables.py:166-181``) and uses
them to validate solver signatures (``crt1d/solvers/__init__.py:32,40,95-109``).  Only the name
lists matter for the hot path; the documentation-generation half of ``variables.py`` is out of scope.
"""

# variables.yml order
CANOPY_RAD_STATE_INPUT_KEYS = [
    "psi", "I_dr0_all", "I_df0_all", "lai", "clump", "leaf_t", "leaf_r", "soil_r", "K_b", "K_b_fn", "G", "G_fn", "mla",
]
SCHEME_OUTPUT_KEYS = ["I_dr", "I_df_d", "I_df_u", "F"]

SHAPES = {
    "psi": "()", "I_dr0_all": "(n_wl,)", "I_df0_all": "(n_wl,)", "lai": "(n_z,)", "clump": "()", "leaf_t": "(n_wl,)",
    "leaf_r": "(n_wl,)", "soil_r": "(n_wl,)", "K_b": "()", "K_b_fn": "callable", "G": "()", "G_fn": "callable", "mla": "()",
    "I_dr": "(n_z, n_wl)", "I_df_d": "(n_z, n_wl)", "I_df_u": "(n_z, n_wl)", "F": "(n_z, n_wl)",
}


class _Vmd:
    """Minimal stand-in for the reference's ``VMD`` object: ``VMD.intent("in"|"out")`` -> dict keyed by name."""

    def intent(self, s):
        if s == "in":
            return {k: SHAPES[k] for k in CANOPY_RAD_STATE_INPUT_KEYS}
        if s == "out":
            return {k: SHAPES[k] for k in SCHEME_OUTPUT_KEYS}
        raise ValueError("intent must be 'in' or 'out'")


VMD = _Vmd()
