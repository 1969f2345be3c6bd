"""
Argument contract of the solver plugins.

The reference derives these lists from ``crt1d/variables.yml`` (``intent: in`` entries at
:37-170, ``intent: out`` at :175-208) through ``Vmd.intent`` (``crt1d/variables.py:166-181``) and uses
them to validate solver signatures (``crt1d/solvers/__init__.py:32,40,95-109``).  The name lists
drive the hot path; ``OUTPUT_METADATA`` carries the dims / CF long name / units of the variables that the reference's
output dataset holds (``crt1d/model.py:338-447`` via ``VmdEntry.dv_tuple``/``da_attrs``, ``crt1d/variables.py:45-75``; values
from ``variables.yml :175-330``).  The documentation-generation half of ``variables.py`` is out of scope.
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


_E = "W m-2"
_EL = "W (m2 leaf)-1"
_UM = "\u03bcm"
_LAI_UL = "(m2 leaf) (m2 ground area)-1"
# name -> (dims, long_name, units, units_long)
OUTPUT_METADATA = {
    "z": (("z",), "Height above ground", "m", None),
    "zm": (("zm",), "Height above ground", "m", None),
    "wl": (("wl",), "Wavelength", _UM, None),
    "wle": (("wle",), "Wavelength of irradiance band edges", _UM, None),
    "dwl": (("wl",), "Wavelength band width", _UM, None),
    "lai": (("z",), "Leaf area index (cumulative)", "m2 m-2", _LAI_UL),
    "dlai": (("zm",), "Leaf area index in layer", "m2 m-2", _LAI_UL),
    "laim": (("zm",), "Leaf area index (cumulative)", "m2 m-2", _LAI_UL),
    "f_slm": (("zm",), "Sunlit leaf fraction", "1", None),
    "I_dr": (("z", "wl"), "Direct beam irradiance (binned)", _E, None),
    "I_df_d": (("z", "wl"), "Downward diffuse irradiance (binned)", _E, None),
    "I_df_u": (("z", "wl"), "Upward diffuse irradiance (binned)", _E, None),
    "F": (("z", "wl"), "Actinic flux (binned)", _E, None),
    "I_d": (("z", "wl"), "Downward irradiance", _E, None),
    "aI": (("zm", "wl"), "Absorbed irradiance", _E, None),
    "aI_l": (("zm", "wl"), "Absorbed irradiance", _E, _EL),
    "aI_dr": (("zm", "wl"), "Absorbed direct irradiance", _E, None),
    "aI_df": (("zm", "wl"), "Absorbed diffuse irradiance", _E, None),
    "aI_sl": (("zm", "wl"), "Absorbed irradiance by sunlit leaves", _E, None),
    "aI_lsl": (("zm", "wl"), "Absorbed irradiance by sunlit leaves", _E, _EL),
    "aI_sh": (("zm", "wl"), "Absorbed irradiance by shaded leaves", _E, None),
    "aI_lsh": (("zm", "wl"), "Absorbed irradiance by shaded leaves", _E, _EL),
    "aI_df_sl": (("zm", "wl"), "Absorbed diffuse irradiance by sunlit leaves", _E, None),
    "aI_df_lsl": (("zm", "wl"), "Absorbed irradiance by sunlit leaves", _E, _EL),
    "aI_df_sh": (("zm", "wl"), "Absorbed diffuse irradiance by shaded leaves", _E, None),
    "aI_df_lsh": (("zm", "wl"), "Absorbed irradiance by shaded leaves", _E, _EL),
    "psi": ((), "Solar zenith angle", "radians", None),
    "sza": ((), "Solar zenith angle", "deg", None),
    "G": ((), "Fractional leaf area in the psi direction", "", None),
    "K_b": ((), "Black leaf attenuation coefficient", "", None),
}


def da_attrs(name):
    """Attributes of one output variable (``VmdEntry.da_attrs``, ``crt1d/variables.py:45-60``)."""
    _, ln, units, ul = OUTPUT_METADATA[name]
    attrs = {"long_name": ln, "units": units}
    if ul:
        attrs["units_long"] = ul
    return attrs


def dv_tuple(name, data):
    """``(dims, data, attrs)`` as ``xarray.Dataset`` takes it (``VmdEntry.dv_tuple``, ``crt1d/variables.py:62-75``)."""
    return (OUTPUT_METADATA[name][0], data, da_attrs(name))
