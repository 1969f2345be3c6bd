"""Band definitions and band-integration weights (host side; these are O(n_wl) and feed the device epilogue).

Mirrors ``crt1d/spectra.py:22-27`` (``BAND_DEFNS_UM``) and ``:71-126`` (``_x_frac_in_bounds``)."""

import warnings

import numpy as np

BAND_DEFNS_UM = {"PAR": (0.4, 0.7), "NIR": (0.7, 2.5), "UV": (0.01, 0.4), "solar": (0.3, 5.0)}


def x_frac_in_bounds(xe, bounds):
    """Fraction of each bin ``[xe[i], xe[i+1]]`` inside ``bounds`` (weights for summing in-band irradiances)."""
    xe = np.asarray(xe, dtype=float)
    x1, x2 = xe[:-1], xe[1:]
    b1, b2 = bounds[0], bounds[1]
    if (b1 < x1[0] or b2 > x2[-1]) and tuple(bounds) != BAND_DEFNS_UM["solar"]:
        warnings.warn(
            f"`bounds` ({b1:.3g}, {b2:.3g}) extend outside the data range defined by `xe` ({x1[0]:.3g}, {x2[-1]:.3g})"
        )
    inside = (x2 >= b1) & (x1 <= b2)
    dx = x2 - x1
    w = np.ones_like(x1)
    left = x1 < b1
    right = ~left & (x2 > b2)
    w[left] = (x2[left] - b1) / dx[left]
    w[right] = (b2 - x1[right]) / dx[right]
    w[~inside] = 0.0
    return w


def band_weights(wle, names=("PAR", "NIR", "solar")):
    """Stack of weight vectors ``(len(names), n_wl)`` for the device band-sum epilogue."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return np.stack([x_frac_in_bounds(wle, BAND_DEFNS_UM[n]) for n in names])


def edges_from_centers_widths(wl, dwl):
    """``wle`` as ``Model._check_inputs`` builds it (``crt1d/model.py:286-287``)."""
    wl, dwl = np.asarray(wl, dtype=float), np.asarray(dwl, dtype=float)
    return np.r_[wl[0] - 0.5 * dwl[0], wl + 0.5 * dwl]
