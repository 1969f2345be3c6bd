"""Spectral band integration of a run's dataset: the computational half of ``crt1d/diagnostics.py`` (``band`` ``:39-108``,
``_E_to_PFD_da`` ``:19-36``; the plotting half is presentation and out of scope).

``band(ds)`` reduces every spectral irradiance variable of a :class:`crt1d_amd.model.Dataset` (``Model.to_dataset()``) over
wavelength with the fractional-overlap weights of the band -- on the device (``crt_hip_band_reduce_f64``), like everything else on the
path.  For batches of columns use ``batched.absorb_bandsum(..., profiles=True)`` / ``IntegratedPlan(..., profiles=True)``, which form
the same sums for up to four band groups while the profiles stream past once (or are never written at all)."""

import warnings

import numpy as np

from .spectra import BAND_DEFNS_UM, e_wl_umol, x_frac_in_bounds


def band_reduce(x, weights):
    """``out[..., g] = sum_wl weights[g, wl] * x[..., wl]`` on the GPU; ``x`` host array ``(..., n_wl)``, ``weights (ngroup <= 4, n_wl)``."""
    import torch

    from . import _lib

    lib = _lib.load()
    if not torch.cuda.is_available():
        raise RuntimeError("crt1d_amd has no CPU path: band integration runs on the GPU")
    x = np.ascontiguousarray(x, dtype=np.float64)
    w = np.ascontiguousarray(np.atleast_2d(weights), dtype=np.float64)
    nb = x.shape[-1]
    if w.shape[1] != nb or not 1 <= w.shape[0] <= 4:
        raise ValueError(f"weights must be (ngroup <= 4, {nb})")
    dev = torch.device("cuda", torch.cuda.current_device())
    xt, wt = torch.as_tensor(x).to(dev).reshape(-1, nb), torch.as_tensor(w).to(dev)
    out = torch.empty((xt.shape[0], w.shape[0]), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        st = lib.crt_hip_band_reduce_f64(xt.data_ptr(), xt.shape[0], nb, wt.data_ptr(), w.shape[0], out.data_ptr(),
                                         torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(st, "crt_hip_band_reduce_f64")
    return out.cpu().numpy().reshape(x.shape[:-1] + (w.shape[0],))


def band(ds, *, variables=None, band_name="PAR", bounds=None, calc_PFD=False):
    """Reduce the spectral variables of ``ds`` by summing in-band irradiances (``diagnostics.py:39-108``): a new
    :class:`~crt1d_amd.model.Dataset` without the wavelength dimension.  ``variables=None``: every variable named ``F`` or containing
    ``I`` that has a ``wl`` dimension (``:84``).  ``calc_PFD=True`` adds photon-flux-density variants (``I`` -> ``PFD`` in the name),
    converted band by band BEFORE summing (``:92-104``): W m-2 / (J per micromole of photons at ``wl``)."""
    from .model import Dataset

    if bounds is None:
        bounds = BAND_DEFNS_UM[band_name]
    wl = np.asarray(ds["wl"], dtype=float)
    if "wle" in ds:
        wle = np.asarray(ds["wle"], dtype=float)
    else:
        warnings.warn("`wle` was not present so we are computing the wave band edges from the band centers (`wl`) and band widths (`dwl`).")
        dwl = np.asarray(ds["dwl"], dtype=float)
        wle = np.r_[wl[0] - 0.5 * dwl[0], wl + 0.5 * dwl]
    w = x_frac_in_bounds(wle, bounds)  # weights as a function of wavelength (:71)
    weights = np.stack([w, w / e_wl_umol(wl)]) if calc_PFD else w[None, :]
    if variables is None:
        variables = [vn for vn, (dims, _, _) in ds.data_vars.items() if (vn == "F" or "I" in vn) and "wl" in dims]
    new = {}
    for vn, (dims, arr, attrs) in ds.data_vars.items():
        if "wl" not in dims:
            new[vn] = (dims, arr, attrs)
            continue
        if vn not in variables:
            continue  # (a spectral variable that is not reduced leaves with the wl dimension)
        if dims[-1] != "wl":
            raise ValueError(f"{vn}: the wavelength axis must be the last one")
        red = band_reduce(arr, weights)
        a = dict(attrs)
        a["long_name"] = f"{attrs.get('long_name', vn)} – {band_name}"
        new[vn] = (dims[:-1], red[..., 0], a)
        if calc_PFD:
            if attrs.get("units") != "W m-2":
                raise AssertionError(f"{vn}: PFD conversion expects W m-2")  # :23
            ap = dict(attrs)
            ap["long_name"] = f"{attrs.get('long_name', vn).replace('irradiance', 'PFD')} – {band_name}"
            ap["units"] = "μmol photons m-2 s-1"
            new[vn.replace("I", "PFD")] = (dims[:-1], red[..., 1], ap)
    coords = {k: v for k, v in ds.coords.items() if k != "wl"}
    keep = set(coords)
    new = {k: v for k, v in new.items() if all(d in keep for d in v[0])}  # (dwl and friends leave with wl)
    attrs = dict(ds.attrs, band_name=band_name, band_bounds=tuple(bounds))
    return Dataset(coords, new, attrs)
