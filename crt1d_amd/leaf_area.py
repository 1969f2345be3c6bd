"""Leaf-area profiles on the device: ``crt1d/leaf_area.py:42-93`` ``distribute_lai_beta`` for many
canopies at once, so that a million-column run builds its ``lai (ncol, nz)`` input in HBM instead of staging it from NumPy."""

from collections import namedtuple

import numpy as np

LeafAreaProfile = namedtuple("LeafAreaProfile", "lai lad z")


def distribute_lai_beta_batched(h_c, LAI, n, *, h_min=0.5, want_lad=True):
    """``h_c``, ``LAI`` (and optionally ``h_min``): ``(ncol,)`` arrays or CUDA tensors.  Returns ``LeafAreaProfile`` of
    ``(ncol, n)`` CUDA float64 tensors (``lad`` is ``None`` with ``want_lad=False``); index 0 = canopy bottom (z = h_min,
    lai = LAI), index n-1 = canopy top (z = h_c, lai = 0), as in the reference."""
    import torch

    from . import _lib

    lib = _lib.load()
    dev = h_c.device if isinstance(h_c, torch.Tensor) and h_c.is_cuda else torch.device("cuda", torch.cuda.current_device())

    def dv(t):
        return torch.as_tensor(t, dtype=torch.float64).to(dev).reshape(-1).contiguous()

    h_c, LAI = dv(h_c), dv(LAI)
    ncol = h_c.numel()
    if LAI.numel() != ncol:
        raise ValueError("h_c and LAI must have the same length")
    if np.isscalar(h_min):
        hm = None if float(h_min) == 0.5 else torch.full((ncol,), float(h_min), dtype=torch.float64, device=dev)
    else:
        hm = dv(h_min)
        if hm.numel() != ncol:
            raise ValueError("h_min must be a scalar or have one value per column")
    if n < 2:
        raise ValueError("need at least two levels")
    lai = torch.empty((ncol, n), dtype=torch.float64, device=dev)
    z = torch.empty_like(lai)
    lad = torch.empty_like(lai) if want_lad else None
    with torch.cuda.device(dev):
        st = lib.crt_hip_lai_beta_f64(h_c.data_ptr(), LAI.data_ptr(), hm.data_ptr() if hm is not None else None, ncol, n, lai.data_ptr(),
                                      z.data_ptr(), lad.data_ptr() if lad is not None else None, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(st, "crt_hip_lai_beta_f64")
    return LeafAreaProfile(lai, lad, z)


def distribute_lai_beta(h_c, LAI, n, *, h_min=0.5):
    """Drop-in for ``crt1d.leaf_area.distribute_lai_beta`` (``crt1d/leaf_area.py:42-93``): one canopy, NumPy arrays out."""
    r = distribute_lai_beta_batched(np.array([float(h_c)]), np.array([float(LAI)]), int(n), h_min=float(h_min))
    return LeafAreaProfile(*(t.cpu().numpy()[0] for t in r))
