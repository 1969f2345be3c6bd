"""
Canopy description: which keys a case must supply and everything that is derived from them before a solve.

The reference keeps this inside ``Model._check_inputs`` (``crt1d/model.py:222-294``) and mutates the parameter dict as it goes.  Here it
is a pure function: :func:`derive` validates a case and returns the derived entries (``lai_tot, lai_eff, dlai, dlai_eff, zm, dz, mu,
wle, K_b_fn, G, K_b``) without touching its argument, so a failed update never leaves a half-written case behind and the same function
serves the single-canopy ``Model`` and host-side preparation of column batches.
"""

import warnings
from collections import namedtuple

import numpy as np

from .solvers.common import KbFunction

# what the reference names the "canopy description" (model.py:21-46); the middle three are derived, the rest are inputs
CANOPY_DESCRIPTION_KEYS = ("lai", "z", "dlai", "lai_tot", "lai_eff", "mla", "clump", "leaf_t", "leaf_r", "soil_r", "wl_leafsoil", "orient", "G_fn")
DERIVED_DESCRIPTION_KEYS = ("dlai", "lai_tot", "lai_eff")
TOC_KEYS = ("I_dr0_all", "I_df0_all", "wl", "dwl", "psi")
INPUT_KEYS = tuple(k for k in CANOPY_DESCRIPTION_KEYS if k not in DERIVED_DESCRIPTION_KEYS) + TOC_KEYS

CanopyDescription = namedtuple("CanopyDescription", CANOPY_DESCRIPTION_KEYS)


class CanopyInputError(AssertionError, ValueError):
    """A case that cannot be solved (the reference asserts; both ``AssertionError`` and ``ValueError`` handlers catch this)."""


def _require(ok, what):
    if not ok:
        raise CanopyInputError(what)


def grid(lai, z, clump):
    """Level grid -> layer quantities.  Index 0 is the ground: heights increase, cumulative LAI decreases to exactly 0 at the top."""
    lai, z = np.asarray(lai, dtype=float), np.asarray(z, dtype=float)
    _require(lai.ndim == 1 and lai.shape == z.shape, f"`lai` {lai.shape} and `z` {z.shape} must be 1-d and of one size")
    _require(z[-1] > z[0], "`z` must increase (index 0 = ground)")
    _require(lai[0] > lai[-1], "cumulative `lai` must decrease with height")
    _require(lai[-1] == 0, "cumulative `lai` must be 0 at the canopy top")
    thickness = np.diff(z)
    per_layer = -np.diff(lai)
    return {"lai_tot": lai[0], "lai_eff": lai * clump, "dlai": per_layer, "dlai_eff": per_layer * clump,
            "zm": z[:-1] + 0.5 * thickness, "dz": thickness}


def band_edges(wl, dwl):
    """Edges of contiguous bands given centres and widths (n + 1 values)."""
    wl, dwl = np.asarray(wl), np.asarray(dwl)
    _require(wl.size == dwl.size, f"`wl` ({wl.size}) and `dwl` ({dwl.size}) differ in size")
    return np.concatenate(([wl[0] - 0.5 * dwl[0]], wl + 0.5 * dwl))


def derive(p):
    """Validate the case ``p`` (a mapping holding :data:`INPUT_KEYS`) and return the dict of derived entries."""
    absent = [k for k in INPUT_KEYS if k not in p]
    if absent:
        raise CanopyInputError(f"required key {absent[0]} is not present. Set it using `update_p`.")
    d = grid(p["lai"], p["z"], p["clump"])
    psi = p["psi"]
    d["mu"] = np.cos(psi)
    if "mu" in p and p["mu"] != d["mu"]:
        warnings.warn("Provided `mu` not consistent with provided `psi`. `mu` will be updated based on the value of `psi`.")
    toc, optical = np.asarray(p["wl"]), np.asarray(p["wl_leafsoil"])
    _require(toc.size == optical.size, f"`wl` ({toc.size}) and `wl_leafsoil` ({optical.size}) differ in size")
    if not np.allclose(toc, optical):
        warnings.warn("Provided wavelengths for optical props (`wl_leafsoil`) and toc BC (`wl`) appear to be incompatible:\n"
                      f"`wl - wl_leafsoil`:\n{toc - optical}")
    d["wle"] = band_edges(p["wl"], p["dwl"])
    G_fn = p["G_fn"]
    d["K_b_fn"] = KbFunction(G_fn)  # psi -> G(psi) / cos(psi); recognisable by the device path, unlike a lambda
    d["G"] = G_fn(psi)
    d["K_b"] = d["K_b_fn"](psi)
    return d


def sizes(p):
    """(number of levels, number of bands) of a case."""
    return np.asarray(p["lai"]).size, np.asarray(p["wl"]).size
