"""
Leaf-angle projection functions G(psi) for the batched canopy-RT path.

Mirrors the ``G_*`` family of the reference (``crt1d/leaf_angle.py:118-202``) but as
*descriptors*: a :class:`GFunction` is callable like the reference's plain functions
(``G_fn(psi)``) and additionally carries ``kind``/``param`` so the HIP column-precompute
kernel can evaluate the same closed form on device instead of receiving a table.

Arbitrary Python callables are still accepted everywhere a ``G_fn`` is expected: the host
wrapper samples them at the library's fixed quadrature nodes (kind ``G_TABLE``).

Device kind ids (must match ``include/crt1d_hip.h``):

== =========================== ==============================================
0  ``G_HORIZONTAL``            cos(psi)                      (ref :118-120)
1  ``G_SPHERICAL``             0.5                           (ref :123-125)
2  ``G_VERTICAL``              2/pi sin(psi)                 (ref :128-130)
3  ``G_ELLIPSOIDAL``           Campbell (1986) exact, x      (ref :133-165)
4  ``G_ELLIPSOIDAL_APPROX``    Campbell (1990) approx, x     (ref :168-180)
5  ``G_ELLIPSOIDAL_APPROX_BONAN`` Ross-Goudriaan, chi_l      (ref :183-202)
6  ``G_TABLE``                 host-sampled callable
== =========================== ==============================================
"""

import math

import numpy as np

G_HORIZONTAL = 0
G_SPHERICAL = 1
G_VERTICAL = 2
G_ELLIPSOIDAL = 3
G_ELLIPSOIDAL_APPROX = 4
G_ELLIPSOIDAL_APPROX_BONAN = 5
G_TABLE = 6

KIND_NAMES = {
    G_HORIZONTAL: "horizontal",
    G_SPHERICAL: "spherical",
    G_VERTICAL: "vertical",
    G_ELLIPSOIDAL: "ellipsoidal",
    G_ELLIPSOIDAL_APPROX: "ellipsoidal_approx",
    G_ELLIPSOIDAL_APPROX_BONAN: "ellipsoidal_approx_bonan",
    G_TABLE: "table",
}


def _ellipsoidal_denominator(x):
    """Campbell (1986) eq. 6 surface-area ratio term (ref ``leaf_angle.py:155-162``)."""
    x = np.asarray(x, dtype=np.float64)
    out = np.full(x.shape, 2.0)
    gt = x > 1
    lt = x < 1
    if np.any(gt):
        e1 = np.sqrt(1 - x[gt] ** -2)
        out[gt] = x[gt] + np.log((1 + e1) / (1 - e1)) / (2 * e1 * x[gt])
    if np.any(lt):
        e2 = np.sqrt(1 - x[lt] ** 2)
        out[lt] = x[lt] + np.arcsin(e2) / e2
    return out


def eval_G(kind, param, psi):
    """Vectorised G(psi) for parameterised kinds; ``kind``/``param``/``psi`` broadcast.

    Uses the tan-free form sqrt(x^2 cos^2 + sin^2)/p2, algebraically equal to the
    reference's sqrt(x^2 + tan^2)/p2 * cos but finite at psi = pi/2.
    """
    kind = np.asarray(kind)
    param = np.asarray(param, dtype=np.float64)
    psi = np.asarray(psi, dtype=np.float64)
    kind, param, psi = np.broadcast_arrays(kind, param, psi)
    out = np.empty(psi.shape, dtype=np.float64)
    c = np.cos(psi)
    s = np.sin(psi)
    m = kind == G_HORIZONTAL
    out[m] = c[m]
    m = kind == G_SPHERICAL
    out[m] = 0.5
    m = kind == G_VERTICAL
    out[m] = 2 / math.pi * s[m]
    m = kind == G_ELLIPSOIDAL
    if np.any(m):
        x = param[m]
        g = np.sqrt(x * x * c[m] ** 2 + s[m] ** 2) / _ellipsoidal_denominator(x)
        out[m] = np.where(x == 1, 0.5, g)
    m = kind == G_ELLIPSOIDAL_APPROX
    if np.any(m):
        x = param[m]
        p2 = x + 1.774 * (x + 1.182) ** -0.733
        out[m] = np.sqrt(x * x * c[m] ** 2 + s[m] ** 2) / p2
    m = kind == G_ELLIPSOIDAL_APPROX_BONAN
    if np.any(m):
        chil = np.clip(param[m], -0.4, 0.6)
        phi1 = 0.5 - 0.633 * chil - 0.330 * chil**2
        phi2 = 0.877 * (1 - 2 * phi1)
        out[m] = phi1 + phi2 * c[m]
    if np.any(kind == G_TABLE):
        raise ValueError("G_TABLE columns have no closed form; sample the callable instead")
    return out


class GFunction:
    """Callable G(psi) that also tells the device which closed form it is."""

    __slots__ = ("kind", "param")

    def __init__(self, kind, param=0.0):
        if kind not in KIND_NAMES or kind == G_TABLE:
            raise ValueError(f"invalid parameterised G kind {kind!r}")
        self.kind = int(kind)
        self.param = float(param)

    def __call__(self, psi):
        res = eval_G(self.kind, self.param, psi)
        return float(res) if res.ndim == 0 else res

    def __repr__(self):
        return f"GFunction({KIND_NAMES[self.kind]}, param={self.param!r})"


def G_horizontal(psi):
    return GFunction(G_HORIZONTAL)(psi)


def G_spherical(psi):
    return GFunction(G_SPHERICAL)(psi)


def G_vertical(psi):
    return GFunction(G_VERTICAL)(psi)


def G_ellipsoidal(psi, x):
    return GFunction(G_ELLIPSOIDAL, x)(psi)


def G_ellipsoidal_approx(psi, x):
    return GFunction(G_ELLIPSOIDAL_APPROX, x)(psi)


def G_ellipsoidal_approx_bonan(psi, xl):
    return GFunction(G_ELLIPSOIDAL_APPROX_BONAN, xl)(psi)


# tag the plain functions so wrappers can recognise the parameter-free ones by identity
G_horizontal.gfunction = GFunction(G_HORIZONTAL)
G_spherical.gfunction = GFunction(G_SPHERICAL)
G_vertical.gfunction = GFunction(G_VERTICAL)


def mla_to_x_approx(mla):
    """Mean leaf angle (deg) -> ellipsoidal x; Campbell (1990) eq. 16 inverted
    (ref ``leaf_angle.py:222-229``). Vectorised."""
    x = (np.deg2rad(mla) / 9.65) ** (-1.0 / 1.65) - 3.0
    if np.any(np.asarray(x) <= 0):
        raise AssertionError("x > 0 required")
    return x


def x_to_mla_approx(x):
    """Ellipsoidal x -> mean leaf angle (deg); ref ``leaf_angle.py:205-211``."""
    return np.rad2deg(9.65 * (3 + np.asarray(x, dtype=np.float64)) ** (-1.65))


def describe_G(G_fn):
    """Return ``(kind, param)`` for a :class:`GFunction`-like ``G_fn``, else ``None``."""
    g = getattr(G_fn, "gfunction", G_fn)
    if isinstance(g, GFunction):
        return g.kind, g.param
    return None
