"""
crt1d_amd -- MI355X (gfx950) native batched 1-D canopy radiative transfer.

A drop-in for the solver path of zmoon/crt1d (``crt1d.solvers.solve_*`` and ``Model.run``):
the per-(column, band) layer solves run in hand-written HIP kernels reached through the C ABI of
``libcrt1d_hip.so`` (``include/crt1d_hip.h``).  There is no CPU fallback.
"""

__version__ = "0.1.0"

from . import leaf_angle  # noqa: F401
