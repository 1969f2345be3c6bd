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


# CODATA 2018 exact SI values (what scipy.constants holds; crt1d/spectra.py:10-13 imports them from there)
_H_PLANCK = 6.62607015e-34  # J s
_C_LIGHT = 299792458.0  # m s-1
_N_AVOGADRO = 6.02214076e23  # mol-1


def e_wl_umol(wl_um):
    """J per micromole of photons at wavelength ``wl_um`` (micrometres); ``crt1d/spectra.py:30-39`` (same operation order)."""
    wl = np.asarray(wl_um, dtype=float) * 1e-6
    e_wl_1 = _H_PLANCK * _C_LIGHT / wl
    e_wl_mol = e_wl_1 * _N_AVOGADRO
    return e_wl_mol * 1e-6


def band_weights(wle, names=("PAR", "NIR", "solar"), *, wl=None, pfd=False):
    """Stack of weight vectors ``(len(names), n_wl)`` for the device band-sum epilogue: ``_x_frac_in_bounds`` of each named band
    (``diagnostics.py:71``).  ``pfd=True``: the photon-flux-density variants of ``diagnostics.band(..., calc_PFD=True)``
    (``:92-104``; ``_E_to_PFD_da`` ``:19-36`` multiplies every band by ``1 / e_wl_umol(wl)`` BEFORE the band sum, because photon energy
    depends on wavelength) -- the same sums with the weights divided by ``e_wl_umol(wl)``; ``wl`` = band centres (micrometres), by
    default the mid-points of ``wle``.  W m-2 in, micromol photons m-2 s-1 out."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        w = np.stack([x_frac_in_bounds(wle, BAND_DEFNS_UM[n]) for n in names])
    if pfd:
        wle = np.asarray(wle, dtype=float)
        wl = 0.5 * (wle[:-1] + wle[1:]) if wl is None else np.asarray(wl, dtype=float)
        w = w * (1.0 / e_wl_umol(wl))[None, :]
    return w


def edges_from_centers_widths(wl, dwl):
    """``wle`` as ``Model._check_inputs`` builds it (``crt1d/model.py:286-287``)."""
    wl, dwl = np.asarray(wl, dtype=float), np.asarray(dwl, dtype=float)
    return np.r_[wl[0] - 0.5 * dwl[0], wl + 0.5 * dwl]


def smear_tuv_batched(x, y, bins, *, out=None):
    """Re-bin ``nspec`` spectra at once on the GPU (``crt_hip_smear_tuv_f64``).

    ``x``: increasing grid ``(nx,)`` shared by all spectra, or ``(nspec, nx)``; ``y``: ``(nspec, nx)``; ``bins``: ``(nbins+1,)``
    edges.  CUDA float64 tensors (host arrays are copied over).  Returns a ``(nspec, nbins)`` CUDA tensor of in-bin
    averages, each the reference's ``_smear_tuv_1`` (``crt1d/spectra.py:221-257``) with identical arithmetic."""
    import ctypes  # noqa: F401
    import torch

    from . import _lib

    lib = _lib.load()
    dev = y.device if isinstance(y, torch.Tensor) and y.is_cuda else torch.device("cuda", torch.cuda.current_device())

    def dv(t):
        return torch.as_tensor(t, dtype=torch.float64).to(dev).contiguous()

    x, y, bins = dv(x), dv(y), dv(bins)
    if y.ndim != 2 or bins.ndim != 1 or bins.numel() < 1:
        raise ValueError("y must be (nspec, nx) and bins (nbins+1,)")
    nspec, nx = y.shape
    if x.shape == (nx,):
        x_stride = 0
    elif x.shape == (nspec, nx):
        x_stride = nx
    else:
        raise ValueError(f"x must be ({nx},) or ({nspec}, {nx}); got {tuple(x.shape)}")
    nbins = bins.numel() - 1
    if out is None:
        out = torch.empty((nspec, nbins), dtype=torch.float64, device=dev)
    elif out.shape != (nspec, nbins) or out.dtype != torch.float64 or not out.is_contiguous():
        raise ValueError("out must be a contiguous float64 (nspec, nbins) tensor")
    if nspec == 0 or nbins == 0:
        return out
    with torch.cuda.device(dev):
        st = lib.crt_hip_smear_tuv_f64(x.data_ptr(), x_stride, nx, y.data_ptr(), nspec, bins.data_ptr(), nbins, out.data_ptr(),
                                       torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(st, "crt_hip_smear_tuv_f64")
    return out


def smear_tuv(x, y, bins):
    """Drop-in for ``crt1d.spectra.smear_tuv`` (``crt1d/spectra.py:260-300``): one spectrum, host arrays in and out."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    return smear_tuv_batched(x, y[None, :], np.asarray(bins, dtype=np.float64)).cpu().numpy()[0]
