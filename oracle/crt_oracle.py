"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU (NumPy) restatement of the zmoon/crt1d canopy-RT solver algorithms, batched over
(column, band).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; ``crt1d_amd`` never does.

Parity status: **pinned** -- every function here is checked (``tests/test_oracle_golden.py``)
against golden vectors produced by running the real reference solvers in the build container
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``; SciPy 1.15.3 / NumPy 2.2.6, versions
recorded in each fixture).  The reference's own test for this path
(``tests/test_n79.py``) needs a network download and could not be run; its *inputs*
(``tests/test_n79.py:13-44``) are one of the golden cases.

Written from the maths of the cited reference lines, vectorised; it is not a copy of the
reference's per-band Python loops.  Conventions (as the reference): ``lai`` is cumulative from
the top, index 0 = ground (total LAI), index -1 = canopy top (0); outputs are
``(ncol, nz, nb)`` with bands contiguous.

Third-party arithmetic the reference delegates to SciPy and how it is restated here:

* ``scipy.integrate.quad`` (QUADPACK, ``common.py:37``, ``_solve_2s.py:32``,
  ``_solve_4s.py:148-149``): fixed composite Gauss-Legendre, graded toward psi = pi/2
  (agrees with mpmath to <= 3e-13; the reference's own quad error is up to ~3e-8).
  ``exact_quad=True`` switches to the very same ``scipy.integrate.quad`` calls.
* ``scipy.sparse.linalg.spsolve`` (``_solve_zq.py:164``): Thomas algorithm (same tridiagonal
  system; agrees to ~1e-13).
* ``scipy.integrate.solve_bvp(tol=1e-6)`` (``_solve_4s.py:241,260``): the linear
  constant-coefficient BVP is solved exactly by eigen-decomposition
  (``solve_4s(method="eig")``); ``method="bvp"`` re-runs SciPy's collocation solver on the
  restated system for like-for-like comparison with the stock reference.
"""

import math

import numpy as np
from numpy.polynomial.legendre import leggauss

PI = math.pi

# ----------------------------------------------------------------------------------------------
# G(psi): crt1d/leaf_angle.py:118-202
# kind ids as in include/crt1d_hip.h (restated here so the oracle imports nothing from the product)
G_HORIZONTAL, G_SPHERICAL, G_VERTICAL, G_ELLIPSOIDAL, G_ELLIPSOIDAL_APPROX, G_ELLIPSOIDAL_APPROX_BONAN = range(6)


def _G_closed_form(kind, param, psi):
    """G for one column's (kind, param) at array ``psi``; leaf_angle.py:118-202."""
    c, s = np.cos(psi), np.sin(psi)
    if kind == G_HORIZONTAL:  # :118-120
        return c + 0.0
    if kind == G_SPHERICAL:  # :123-125
        return np.full_like(psi, 0.5)
    if kind == G_VERTICAL:  # :128-130
        return 2 / PI * s
    if kind == G_ELLIPSOIDAL:  # :133-165 (x == 1 special-cased :147-149)
        x = param
        if x == 1:
            return np.full_like(psi, 0.5)
        if x > 1:
            e = math.sqrt(1 - x**-2)
            p2 = x + math.log((1 + e) / (1 - e)) / (2 * e * x)
        else:
            e = math.sqrt(1 - x**2)
            p2 = x + math.asin(e) / e
        return np.sqrt(x * x * c * c + s * s) / p2  # == sqrt(x^2+tan^2)/p2 * cos
    if kind == G_ELLIPSOIDAL_APPROX:  # :168-180
        x = param
        p2 = x + 1.774 * (x + 1.182) ** -0.733
        return np.sqrt(x * x * c * c + s * s) / p2
    if kind == G_ELLIPSOIDAL_APPROX_BONAN:  # :183-202
        chil = min(max(param, -0.4), 0.6)
        phi1 = 0.5 - 0.633 * chil - 0.330 * chil**2
        phi2 = 0.877 * (1 - 2 * phi1)
        return phi1 + phi2 * c
    raise ValueError(f"unknown G kind {kind}")


class Columns:
    """Per-column geometry: psi (ncol,), lai (ncol,nz), mla (ncol,) and G description."""

    def __init__(self, psi, lai, *, mla=None, g_kind=None, g_param=None, G_fn=None):
        self.psi = np.atleast_1d(np.asarray(psi, dtype=np.float64))
        self.lai = np.atleast_2d(np.asarray(lai, dtype=np.float64))
        self.ncol, self.nz = self.lai.shape
        assert self.psi.shape == (self.ncol,)
        self.mla = None if mla is None else np.broadcast_to(np.asarray(mla, dtype=np.float64), (self.ncol,))
        self.G_fn = G_fn
        if G_fn is None:
            self.g_kind = np.broadcast_to(np.asarray(g_kind), (self.ncol,))
            self.g_param = np.broadcast_to(np.asarray(g_param, dtype=np.float64), (self.ncol,))

    def G(self, c, psi):
        """G of column ``c`` at array/scalar psi."""
        psi = np.asarray(psi, dtype=np.float64)
        if self.G_fn is not None:
            if psi.ndim == 0:
                return np.float64(self.G_fn(float(psi)))
            r = self.G_fn(psi)
            r = np.asarray(r, dtype=np.float64)
            if r.shape != psi.shape:  # callables that only handle scalars / return scalars
                r = np.array([self.G_fn(float(p)) for p in psi.ravel()], dtype=np.float64).reshape(psi.shape)
            return r
        return _G_closed_form(int(self.g_kind[c]), float(self.g_param[c]), psi)

    def K_b(self):
        """K_b = G(psi)/cos(psi) per column; model.py:291-293."""
        return np.array([self.G(c, self.psi[c]) for c in range(self.ncol)]) / np.cos(self.psi)


# ----------------------------------------------------------------------------------------------
# quadrature (replaces QUADPACK)


def _graded_rule(n=24, edges=(0.0, 3e-5, 3e-4, 3e-3, 0.03, 0.2, 0.45, 0.7, 1.0)):
    """Composite Gauss-Legendre on psi in [0, pi/2]: panels (edges in t = pi/2 - psi, fractions of pi/2) graded toward pi/2, where
    e^{-G L / cos psi} has its boundary layer, and split at the wide end for 1/G of small-x ellipsoidal distributions.  8 x 24 nodes:
    1 - tau_d(L) and mu_bar to <= 3e-15 relative for L in [1e-4, 20], x in [0.2, 3] against 30-digit quadrature -- deliberately a different
    and finer rule than the device's 6 x 16 (csrc/colpre.hip), so that HIP-vs-oracle parity also checks the device's quadrature."""
    x, w = leggauss(n)
    T = PI / 2
    e = [T * f for f in edges]
    ps, ws = [], []
    for a, b in zip(e[:-1], e[1:]):
        t = a + (x + 1) * (b - a) / 2
        ps.append(T - t)
        ws.append(w * (b - a) / 2)
    return np.concatenate(ps), np.concatenate(ws)


_PSI_Q, _W_Q = _graded_rule()


def tau_d(cols, L, *, method="quad", exact_quad=False):
    """Diffuse transmittance tau_d(L) = 2 int_0^{pi/2} exp(-K_b(psi) L) sin cos dpsi per column.

    ``L`` is ``(ncol, m)``.  common.py:30-37 ('quad'), :40-53 ('9sky'), :56-87 (dispatch;
    ``ValueError`` for other methods, :78).
    """
    L = np.asarray(L, dtype=np.float64)
    out = np.empty_like(L)
    if method == "9sky":
        psis = np.deg2rad(np.arange(5.0, 90.0, 10.0))
        wts = np.sin(psis) * np.cos(psis) * (2 * math.radians(10))
    elif method == "quad":
        psis = _PSI_Q
        wts = 2 * _W_Q * np.sin(psis) * np.cos(psis)
    else:
        raise ValueError("invalid `method`. Valid options are 'quad' and '9sky'.")
    for c in range(cols.ncol):
        if method == "quad" and exact_quad:
            from scipy import integrate

            for j, Lv in enumerate(L[c]):
                f = lambda p: math.exp(-float(cols.G(c, p)) / math.cos(p) * Lv) * math.sin(p) * math.cos(p)  # noqa: E731
                out[c, j] = 2 * integrate.quad(f, 0, PI / 2, epsrel=1e-9)[0]
            continue
        k = cols.G(c, psis) / np.cos(psis)  # (nq,)
        out[c] = np.exp(-np.outer(L[c], k)) @ wts
    return out


def one_minus_tau_d(cols, L, *, method="quad", exact_quad=False):
    """1 - tau_d(L).  For the fixed-node rule it is integrated directly, 2 int (1 - e^{-K_b L}) sin cos dpsi with expm1 (the weights
    integrate to one), which keeps its RELATIVE accuracy as L -> 0; n79 divides it by dlai (_solve_n79.py:146,154-155), and formed as
    1 - tau_d the ~3e-13 by which two quadrature rules differ in tau_d became ~2e-8 at dlai ~ 3e-4.  '9sky' and the reference's own
    adaptive quadrature (exact_quad) stay 1 - tau_d, as the reference computes it (_solve_n79.py:99,146)."""
    L = np.asarray(L, dtype=np.float64)
    if method != "quad" or exact_quad:
        return 1 - tau_d(cols, L, method=method, exact_quad=exact_quad)
    wts = 2 * _W_Q * np.sin(_PSI_Q) * np.cos(_PSI_Q)
    out = np.empty_like(L)
    for c in range(cols.ncol):
        k = cols.G(c, _PSI_Q) / np.cos(_PSI_Q)
        out[c] = -np.expm1(-np.outer(L[c], k)) @ wts
    return out


def mu_bar(cols, *, exact_quad=False):
    """mu_bar = int_0^{pi/2} cos sin / G dpsi (Sellers 1985 p.1336); _solve_2s.py:32."""
    out = np.empty(cols.ncol)
    for c in range(cols.ncol):
        if exact_quad:
            from scipy import integrate

            out[c] = integrate.quad(
                lambda sa: math.cos(sa) / float(cols.G(c, sa)) * -math.sin(sa), PI / 2, 0
            )[0]
        else:
            out[c] = np.sum(_W_Q * np.cos(_PSI_Q) * np.sin(_PSI_Q) / cols.G(c, _PSI_Q))
    return out


def G_integrals(cols, mu_s, *, exact_quad=False):
    """(int_0^{mu_s} G(acos m) dm, int_{mu_s}^1 G(acos m) dm) per column; _solve_4s.py:148-149."""
    x, w = leggauss(24)
    out = np.empty((cols.ncol, 2))
    for c in range(cols.ncol):
        if exact_quad:
            from scipy import integrate

            g = lambda m: float(cols.G(c, math.acos(m)))  # noqa: E731
            out[c, 0] = integrate.quad(g, 0, mu_s)[0]
            out[c, 1] = integrate.quad(g, mu_s, 1)[0]
            continue
        for k, (a, b) in enumerate(((0.0, mu_s), (mu_s, 1.0))):
            # substitute m = cos(psi) so vertical-leaf G (sqrt(1-m^2)) stays smooth
            pa, pb = math.acos(b), math.acos(a)
            p = pa + (x + 1) * (pb - pa) / 2
            out[c, k] = np.sum(w * (pb - pa) / 2 * cols.G(c, p) * np.sin(p))
    return out


# ----------------------------------------------------------------------------------------------
# helpers


def _bc(a, ncol):
    """Per-(column, band) input -> (ncol, 1, nb) for broadcasting against (ncol, nz, 1)."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a[None, :]
    a = np.broadcast_to(a, (ncol, a.shape[-1]))
    return a[:, None, :]


def _thomas(a, b, c, d):
    """Tridiagonal solve along axis 0 (sub a, diag b, super c, rhs d); _solve_n79.py:167-200."""
    n = a.shape[0]
    e = np.empty_like(d)
    f = np.empty_like(d)
    e[0] = c[0] / b[0]
    f[0] = d[0] / b[0]
    for i in range(1, n):
        den = b[i] - a[i] * e[i - 1]
        e[i] = c[i] / den
        f[i] = (d[i] - a[i] * f[i - 1]) / den
    u = np.empty_like(d)
    u[n - 1] = f[n - 1]
    for i in range(n - 2, -1, -1):
        u[i] = f[i] - e[i] * u[i + 1]
    return u


# ----------------------------------------------------------------------------------------------
# schemes


def solve_2s(cols, *, I_dr0, I_df0, leaf_r, leaf_t, soil_r, exact_quad=False):
    """Dickinson-Sellers two-stream closed form; _solve_2s.py:11-163."""
    nc = cols.ncol
    K = cols.K_b()[:, None, None]  # :26,40
    mu = np.cos(cols.psi)[:, None, None]  # :27
    cos2tb = (np.cos(np.deg2rad(cols.mla)) ** 2)[:, None, None]  # :28,68
    mb = mu_bar(cols, exact_quad=exact_quad)[:, None, None]  # :32
    L = cols.lai[:, :, None]  # :37
    LT = cols.lai[:, 0][:, None, None]  # :38
    I_dr0, I_df0 = _bc(I_dr0, nc), _bc(I_df0, nc)
    al, ta, rs = _bc(leaf_r, nc), _bc(leaf_t, nc), _bc(soil_r, nc)

    om = al + ta  # :65
    beta = 0.5 * (al + ta + (al - ta) * cos2tb) / om  # :68
    a_s = om / 2 * (1 - mu * np.log((mu + 1) / mu))  # :73
    beta0 = (1 + mb * K) / (om * mb * K) * a_s  # :76
    b = 1 - (1 - beta) * om  # :80
    c = om * beta
    d = om * mb * K * beta0
    f = om * mb * K * (1 - beta0)
    h = np.sqrt(b * b - c * c) / mb
    sig = (mb * K) ** 2 + c * c - b * b  # :85
    u1 = b - c / rs  # :87
    u2 = b - c * rs
    u3 = f + c * rs
    S1 = np.exp(-h * LT)
    S2 = np.exp(-K * LT)
    p1, p2, p3, p4 = b + mb * h, b - mb * h, b + mb * K, b - mb * K
    D1 = p1 * (u1 - mb * h) / S1 - p2 * (u1 + mb * h) * S1  # :96
    D2 = (u2 + mb * h) / S1 - (u2 - mb * h) * S1
    h1 = -d * p4 - c * f  # :99
    t1 = d - h1 / sig * p3
    t2 = d - c - h1 / sig * (u1 + mb * K)
    h2 = (t1 * (u1 - mb * h) / S1 - p2 * t2 * S2) / D1
    h3 = -(t1 * (u1 + mb * h) * S1 - p1 * t2 * S2) / D1
    h4 = -f * p3 - c * d  # :108 Sellers (1996)
    t3 = u3 - h4 / sig * (u2 - mb * K)
    h5 = -(h4 / sig * (u2 + mb * h) / S1 + t3 * S2) / D2
    h6 = (h4 / sig * (u2 - mb * h) * S1 + t3 * S2) / D2
    h7 = c / D1 * (u1 - mb * h) / S1
    h8 = -c / D1 * (u1 + mb * h) * S1
    h9 = (u2 + mb * h) / S1 / D2
    h10 = -(u2 - mb * h) * S1 / D2  # :120

    eK = np.exp(-K * L)
    em = np.exp(-h * L)
    ep = np.exp(h * L)
    up = I_dr0 * (h1 * eK / sig + h2 * em + h3 * ep) + I_df0 * (h7 * em + h8 * ep)  # :125,130,134
    dn = I_dr0 * (h4 * eK / sig + h5 * em + h6 * ep) + I_df0 * (h9 * em + h10 * ep)  # :126,131,135
    I_dr = I_dr0 * eK  # :150
    return {"I_dr": I_dr, "I_df_d": dn, "I_df_u": up, "F": I_dr / mu + 2 * up + 2 * dn}  # :153-156


def solve_bl(cols, *, I_dr0, I_df0, leaf_r, leaf_t, exact_quad=False):
    """Beer-Lambert; _solve_bl.py:9-93."""
    nc = cols.ncol
    K_b = cols.K_b()[:, None, None]
    mu = np.cos(cols.psi)[:, None, None]
    L = cols.lai[:, :, None]
    tau_b = np.exp(-K_b * L)  # :31
    tau_df = tau_d(cols, cols.lai, exact_quad=exact_quad)[:, :, None]  # :35-37 (every level)
    I_dr0, I_df0 = _bc(I_dr0, nc), _bc(I_df0, nc)
    kp = np.sqrt(1 - (_bc(leaf_t, nc) + _bc(leaf_r, nc)))  # :58-60
    tau_g = np.exp(-(K_b * kp) * L)  # :62-65
    I_dr = I_dr0 * tau_b  # :69
    I_df = I_df0 * tau_df + 0.5 * (I_dr0 * (tau_g - tau_b))  # :70,74,79
    return {"I_dr": I_dr, "I_df_d": I_df, "I_df_u": np.zeros_like(I_df), "F": I_dr / mu + 2 * I_df}  # :85-90


def solve_g77(cols, *, I_dr0, I_df0, leaf_r, leaf_t, soil_r):
    """Goudriaan (1977) per Bodin & Franklin (2012); _solve_g77.py:7-135."""
    nc = cols.ncol
    assert np.all(cols.lai[:, 0] == cols.lai.max(axis=1))  # :35
    kb = cols.K_b()[:, None, None]
    mu = np.cos(cols.psi)[:, None, None]
    L = cols.lai[:, :, None]
    LT = cols.lai[:, 0][:, None, None]
    I_dr0, I_df0 = _bc(I_dr0, nc), _bc(I_df0, nc)
    r_l, t_l, W = _bc(leaf_r, nc), _bc(leaf_t, nc), _bc(soil_r, nc)
    sigma = r_l + t_l  # :57
    kp = np.sqrt(1 - sigma)  # :59
    rho_c = ((1 - kp) / (1 + kp)) * (2 / (1 + 1.6 * mu))  # :66
    k_d = 0.8 * np.sqrt(1 - sigma)  # :69
    I_df = I_df0 * (1 - rho_c) * np.exp(-k_d * L)  # :73
    A_sl = np.exp(-kb * L)  # :80
    I_dr = I_dr0 * A_sl  # :77
    I_sc = I_dr0 * (1 - rho_c) * np.exp(-kp * kb * L) - I_dr0 * (1 - sigma) * A_sl  # :84-86
    I_sc_d = 0.5 * I_sc
    I_sc_u = 0.5 * I_sc
    I_sr = W * (I_dr0 * A_sl[:, :1] + I_df[:, :1] + I_sc_d[:, :1]) * np.exp(-k_d * (LT - L))  # :95
    common = k_d / kp * I_df + k_d / np.sqrt(1 - r_l) * I_sc_u + k_d / np.sqrt(1 - t_l) * I_sc_d
    a_sh = (1 - A_sl) * common  # :99-101
    a_sl = A_sl * (common + kb * I_dr0)  # :106-111
    dn = I_sc_d + I_df  # :115
    up = I_sc_u + I_sr  # :116
    return {
        "I_dr": I_dr, "I_df_d": dn, "I_df_u": up, "F": I_dr / mu + 2 * up + 2 * dn,
        "aI_lsl": a_sl, "aI_lsh": a_sh, "aI_l": a_sl + a_sh,
    }  # :119-135


def solve_bf(cols, *, I_dr0, I_df0, leaf_r, leaf_t, soil_r):
    """Bodin & Franklin (2012) improved Goudriaan; _solve_bf.py:7-154 (scope row f, rank 1).

    As the reference: incoming diffuse is *not* reduced by (1 - rho_c) (:83), and the returned
    ``rho_c`` is the value of the **last band only** (:78,153) -- here one scalar per column.
    """
    nc = cols.ncol
    assert np.all(cols.lai[:, 0] == cols.lai.max(axis=1))  # :39
    kb = cols.K_b()[:, None, None]
    mu = np.cos(cols.psi)[:, None, None]
    L = cols.lai[:, :, None]
    LT = cols.lai[:, 0][:, None, None]
    I_dr0, I_df0 = _bc(I_dr0, nc), _bc(I_df0, nc)
    r_l, t_l, W = _bc(leaf_r, nc), _bc(leaf_t, nc), _bc(soil_r, nc)
    sigma = r_l + t_l
    kp = np.sqrt(1 - sigma)  # :70
    rho_c = ((1 - kp) / (1 + kp)) * (2 / (1 + 1.6 * mu))  # :77
    k_d = 0.8 * np.sqrt(1 - sigma)  # :80
    I_df = I_df0 * np.exp(-k_d * L)  # :84
    A_sl = np.exp(-kb * L)  # :91
    I_dr = I_dr0 * A_sl  # :88
    I_sc_d = I_dr0 * t_l * ((A_sl - np.exp(-k_d * L)) / (k_d - kb))  # B&F eq. 8 :95
    I_sc_u = I_dr0 * r_l * ((A_sl - np.exp(k_d * L - (kb + k_d) * LT)) / (k_d + kb))  # eq. 9 :99-103
    I_sr = W * (I_dr0 * A_sl[:, :1] + I_df[:, :1] + I_sc_d[:, :1]) * np.exp(-k_d * (LT - L))  # :112
    common = k_d / kp * I_df + k_d / np.sqrt(1 - r_l) * I_sc_u + k_d / np.sqrt(1 - t_l) * I_sc_d
    a_sh = (1 - A_sl) * common  # :116-118
    a_sl = A_sl * (common + kb * I_dr0)  # :123-128
    dn = I_sc_d + I_df  # :131
    up = I_sc_u + I_sr  # :132
    return {
        "I_dr": I_dr, "I_df_d": dn, "I_df_u": up, "F": I_dr / mu + 2 * up + 2 * dn,
        "aI_lsl": a_sl, "aI_lsh": a_sh, "aI_l": a_sl + a_sh, "rho_c": rho_c[:, 0, -1],
    }


def solve_n79(cols, *, I_dr0, I_df0, leaf_r, leaf_t, soil_r, tau_d_method="quad", exact_quad=False):
    """Norman (1979) / Bonan SP 14.3; _solve_n79.py:11-164 (quirks of :85-92 kept as-is)."""
    nc, nz = cols.ncol, cols.nz
    K_b = cols.K_b()
    lai = cols.lai
    dlai = lai[:, :-1] - lai[:, 1:]  # :40
    tb = np.exp(-K_b[:, None] * dlai)  # :45
    tbcum = np.exp(-K_b[:, None] * lai)  # :46
    if tau_d_method not in ("quad", "9sky"):
        raise ValueError("invalid `method`. Valid options are 'quad' and '9sky'.")
    omtd = one_minus_tau_d(cols, dlai, method=tau_d_method, exact_quad=exact_quad)  # 1 - td, :53
    td = 1 - omtd
    omtb = -np.expm1(-K_b[:, None] * dlai) if not exact_quad else 1 - tb  # 1 - tb
    laim = (lai[:, :-1] + lai[:, 1:]) / 2  # :57
    fsun = np.exp(-K_b[:, None] * laim)  # :58
    fsha = 1 - fsun

    # everything below: (2nz | nz | nz-1, ncol, nb)
    swb = np.moveaxis(_bc(I_dr0, nc), 1, 0)[0]  # (ncol, nb)
    swd = np.moveaxis(_bc(I_df0, nc), 1, 0)[0]
    rho = np.moveaxis(_bc(leaf_r, nc), 1, 0)[0]
    tau = np.moveaxis(_bc(leaf_t, nc), 1, 0)[0]
    alb = np.moveaxis(_bc(soil_r, nc), 1, 0)[0]
    nb = rho.shape[-1]
    n = 2 * nz
    a = np.zeros((n, nc, nb))
    b = np.ones((n, nc, nb))
    c = np.zeros((n, nc, nb))
    d = np.zeros((n, nc, nb))

    def layer(j):  # scattering coefficients of layer j (uses td[j])
        t, omt = td[:, j][:, None], omtd[:, j][:, None]
        refld = omt * rho
        trand = omt * tau + t
        return refld - trand * trand / refld, trand / refld

    # soil, upward (:79-82)
    c[0] = -alb
    d[0] = swb * tbcum[:, 0][:, None] * alb
    # first downward row uses index **1** of td/tb/tbcum (:85-92)
    aiv, biv = layer(1)
    a[1], c[1] = -aiv, -biv
    d[1] = swb * tbcum[:, 1][:, None] * omtb[:, 1][:, None] * (tau - rho * biv)
    for j in range(nz - 2):  # :95-119
        ju, jd = 2 * (j + 1), 2 * (j + 1) + 1
        fiv, eiv = layer(j)
        a[ju], c[ju] = -eiv, -fiv
        d[ju] = swb * tbcum[:, j + 1][:, None] * omtb[:, j][:, None] * (rho - tau * eiv)
        aiv, biv = layer(j + 1)
        a[jd], c[jd] = -aiv, -biv
        d[jd] = swb * tbcum[:, j + 2][:, None] * omtb[:, j + 1][:, None] * (tau - rho * biv)
    fiv, eiv = layer(nz - 2)  # top layer upward: td[-1] (:122-129)
    a[n - 2], c[n - 2] = -eiv, -fiv
    d[n - 2] = swb * tbcum[:, -1][:, None] * omtb[:, -1][:, None] * (rho - tau * eiv)
    d[n - 1] = swd  # :132-135

    u = _thomas(a, b, c, d)  # :138
    swup = np.moveaxis(u[0::2], 0, 1)  # (ncol, nz, nb) :141
    swdn = np.moveaxis(u[1::2], 0, 1)  # :142

    om = (rho + tau)[:, None, :]
    direct = swb[:, None, :] * tbcum[:, 1:, None] * omtb[:, :, None] * (1 - om)  # :145
    diffuse = (swdn[:, 1:] + swup[:, :-1]) * omtd[:, :, None] * (1 - om)  # :146
    sun = diffuse * fsun[:, :, None] + direct
    shade = diffuse * fsha[:, :, None]
    I_dr = swb[:, None, :] * tbcum[:, :, None]  # :151
    mu = np.cos(cols.psi)[:, None, None]
    return {
        "I_dr": I_dr, "I_df_d": swdn, "I_df_u": swup, "F": I_dr / mu + 2 * swdn + 2 * swup,  # :161
        "aI_lsl": sun / (fsun * dlai)[:, :, None],  # :154
        "aI_lsh": shade / (fsha * dlai)[:, :, None],  # :155
    }


def solve_zq(cols, *, I_dr0, I_df0, leaf_r, leaf_t, soil_r, exact_quad=False):
    """Zhao & Qualls (2005) multi-scatter; _solve_zq.py:13-229. Direct tridiagonal solve."""
    nc, m = cols.ncol, cols.nz
    K = cols.K_b()
    mu = np.cos(cols.psi)
    lai = cols.lai
    dl = np.diff(lai, axis=1)  # :30
    dlm = np.array([abs(np.mean(r[r != 0])) for r in dl])  # :50 single mean dlai per column
    tau_i = tau_d(cols, dlm[:, None], exact_quad=exact_quad)[:, 0]  # :51
    t_psi = np.exp(-K * dlm)  # :52

    S0 = _bc(I_dr0, nc)[:, 0, :]  # (ncol, nb)
    I_df0 = _bc(I_df0, nc)[:, 0, :]
    bL = _bc(leaf_r, nc)[:, 0, :]
    tL = _bc(leaf_t, nc)[:, 0, :]
    rho = _bc(soil_r, nc)[:, 0, :]
    nb = bL.shape[-1]
    aL = 1 - (bL + tL)  # :87
    r_i = 2.0 / 3 * (bL / (bL + tL)) + 1.0 / 3 * (tL / (bL + tL))  # eq. 23 :40-43
    r_psi = 0.5 + 0.3334 * ((bL - tL) / (bL + tL)) * mu[:, None]  # eq. 22 :35-38

    # r, t, a over layer index 0..m+1 (0 = ground, m+1 = top ghost); :101-108
    shp = (m + 2, nc, nb)
    r = np.broadcast_to(r_i, shp).copy()
    t = np.broadcast_to(tau_i[:, None], shp).copy()
    a = np.broadcast_to(aL, shp).copy()
    r[0], r[-1] = 1, 0
    t[0], t[-1] = 0, 1
    a[0], a[-1] = 1 - rho, 0

    n = 2 * m + 2
    sub = np.zeros((n, nc, nb))
    dia = np.zeros((n, nc, nb))
    sup = np.zeros((n, nc, nb))
    C = np.zeros((n, nc, nb))
    li = np.arange(1, m + 1)
    fwd = t[li] + (1 - t[li]) * (1 - a[li]) * (1 - r[li])  # transmitted + forward scattered
    q_lo = r[li - 1] * (1 - a[li - 1]) * (1 - t[li - 1])
    q_me = r[li] * (1 - a[li]) * (1 - t[li])
    q_hi = r[li + 1] * (1 - a[li + 1]) * (1 - t[li + 1])
    dia[0] = 1  # :115
    sub[2 * li - 1] = -fwd  # :116
    dia[2 * li - 1] = -q_lo * fwd  # :117  (note: q_lo carries r[li-1])
    sup[2 * li - 1] = 1 - q_lo * q_me  # :118
    sub[2 * li] = 1 - q_me * q_hi  # :119
    dia[2 * li] = -q_hi * fwd  # :120
    sup[2 * li] = -fwd  # :121
    dia[n - 1] = 1  # :122

    S = S0[:, None, :] * np.exp(-K[:, None] * lai)[:, :, None]  # (ncol, m, nb) :130
    Sm = np.moveaxis(S, 1, 0)  # (m, ncol, nb)
    C[0] = rho * Sm[0]  # :136
    C[2 * li - 1] = (1 - q_lo * q_me) * r_psi * (1 - t_psi[:, None]) * (1 - a[li]) * Sm  # :137-139
    C[2 * li] = (1 - q_me * q_hi) * (1 - t_psi[:, None]) * (1 - a[li]) * (1 - r_psi) * Sm  # :140-142
    C[n - 1] = I_df0  # :143

    x = _thomas(sub, dia, sup, C)  # :158-164
    SWu0, SWd0 = x[0::2], x[1::2]  # (m+1, ncol, nb) :166-167
    den = 1 - q_lo * q_me
    SWd = np.zeros_like(SWd0)
    SWu = np.zeros_like(SWu0)
    SWd[li] = SWd0[li] / den + q_me * SWu0[li - 1] / den  # eq. 24 :180-182
    SWu[li - 1] = SWu0[li - 1] / den + q_lo * SWd0[li] / den  # eq. 25 :185-187

    mv = lambda z: np.moveaxis(z, 0, 1)  # noqa: E731
    mu3 = mu[:, None, None]
    dn_ss, dn = mv(SWd0[1:]), mv(SWd[1:])  # :197-198
    up_ss, up = mv(SWu0[:-1]), mv(SWu[:-1])  # :199-200
    return {
        "I_dr": S, "I_df_d": dn, "I_df_u": up, "F": S / mu3 + 2 * up + 2 * dn,
        "I_df_d_ss": dn_ss, "I_df_u_ss": up_ss, "F_ss": S / mu3 + 2 * up_ss + 2 * dn_ss,
    }  # :201-229


def solve_zq_pa(cols, *, I_dr0, I_df0, leaf_r, leaf_t, soil_r, clump=1.0, exact_quad=False):
    """Zhao & Qualls, pyAPES variant; _solve_zq_pa.py:24-418 (scope row f, rank 3).

    The canopy is regridded to M = min(100, N) equal layers (:94-100), the zq tridiagonal system is solved there
    (:196-278) with tau_d(LAI/M) for every layer (:176), eqs. 24/25 give SWd, SWu on the M+1 interfaces (:286-335)
    with the end values copied from their neighbours (:310,335), and the result is interpolated linearly in cumulative
    LAI back to the original levels (:357-362).  `clump` and K_d only enter absorption terms the reference computes
    but does not return (:364-404)."""
    nc, N = cols.ncol, cols.nz
    M = min(100, N)
    Kb = cols.K_b()
    mu = np.cos(cols.psi)
    LAI = cols.lai[:, 0]
    I_dr0a = _bc(I_dr0, nc)[:, 0, :]
    I_df0a = _bc(I_df0, nc)[:, 0, :]
    bL = _bc(leaf_r, nc)[:, 0, :]
    tauL = _bc(leaf_t, nc)[:, 0, :]
    rho = _bc(soil_r, nc)[:, 0, :]
    nb = bL.shape[-1]
    alb = bL + tauL
    Lm = LAI / M
    taud_i = tau_d(cols, Lm[:, None], exact_quad=exact_quad)[:, 0]  # :176
    taub_i = np.exp(-Kb * Lm)  # :174

    shp = (M + 2, nc, nb)
    aL = np.broadcast_to(1 - alb, shp).copy()  # :141
    tL = np.broadcast_to(tauL / alb, shp).copy()
    rL = np.broadcast_to(bL / alb, shp).copy()
    aL[0], tL[0], rL[0] = 1 - rho, 0, 1  # :148-150
    aL[-1], tL[-1], rL[-1] = 0, 1, 0  # :153-155
    L = np.full((M + 2, nc), 1.0) * Lm[None, :]
    L[0] = 0
    L[-1] = 0
    Lcum = np.cumsum(L[::-1], axis=0)  # from the top: 0, Lm, ..., LAI, LAI   (:159)
    f_sl = np.exp(-Kb[None, :] * Lcum)[::-1]  # :164
    Ib = f_sl[:, :, None] * I_dr0a[None]  # :168
    taub = np.broadcast_to(taub_i[None, :, None], shp).copy()
    taud = np.broadcast_to(taud_i[None, :, None], shp).copy()
    taub[0] = 0
    taud[0] = 0  # :180-181
    # note: taub/taud of the top ghost layer (L = 0) are exp(0) = 1 for taub but tau_d(LAI/M) for taud (:176 fills all)
    taub[-1] = 1.0
    rb = 0.5 + 0.3334 * (rL - tL) / (rL + tL) * mu[None, :, None]  # :186
    rd = 2.0 / 3.0 * rL / (rL + tL) + 1.0 / 3.0 * tL / (rL + tL)  # :187
    rb[0], rd[0], rb[-1], rd[-1] = 1, 1, 0, 0  # :189-192

    n = 2 * M + 2
    sub = np.zeros((n, nc, nb))
    dia = np.zeros((n, nc, nb))
    sup = np.zeros((n, nc, nb))
    C = np.zeros((n, nc, nb))
    k = np.arange(1, M + 1)
    fwd = taud[k] + (1 - taud[k]) * (1 - aL[k]) * (1 - rd[k])
    q = rd * (1 - aL) * (1 - taud)  # r (1-a) (1-t) per layer
    dlo = 1 - q[k - 1] * q[k]
    dhi = 1 - q[k] * q[k + 1]
    dia[0] = 1  # :199
    sub[2 * k - 1] = -fwd  # :207
    dia[2 * k - 1] = -q[k - 1] * fwd  # :208-213
    sup[2 * k - 1] = dlo  # :214-216
    sub[2 * k] = dhi  # :218-220
    dia[2 * k] = -q[k + 1] * fwd  # :221-226
    sup[2 * k] = -fwd  # :227
    dia[n - 1] = 1  # :230
    C[0] = rho * Ib[0]  # :238
    C[2 * k - 1] = dlo * rb[k] * (1 - taub[k]) * (1 - aL[k]) * Ib[k]  # :242-255
    C[2 * k] = dhi * (1 - taub[k]) * (1 - aL[k]) * (1 - rb[k]) * Ib[k]  # :256-269
    C[n - 1] = I_df0a  # :274
    x = _thomas(sub, dia, sup, C)  # np.linalg.solve, :277
    SWu0, SWd0 = x[0::2], x[1::2]
    SWd = np.zeros((M + 1, nc, nb))
    SWu = np.zeros((M + 1, nc, nb))
    kk = np.arange(0, M)
    den = 1 - q[kk] * q[kk + 1]
    SWd[kk + 1] = SWd0[kk + 1] / den + SWu0[kk] * q[kk + 1] / den  # eq. 24 :286-309
    SWd[0] = SWd[1]  # :310
    SWu[kk] = SWu0[kk] / den + SWd0[kk + 1] * q[kk] / den  # eq. 25 :313-334
    SWu[M] = SWu[M - 1]  # :335

    dn = np.empty((nc, N, nb))
    up = np.empty((nc, N, nb))
    for c in range(nc):
        xi = Lcum[: M + 1, c]  # 0 .. LAI ascending = cumulative LAI of interfaces M .. 0   (:338-341,359)
        X = cols.lai[c][::-1]
        for i in range(nb):
            dn[c, :, i] = np.interp(X, xi, SWd[::-1, c, i])[::-1]  # :360
            up[c, :, i] = np.interp(X, xi, SWu[::-1, c, i])[::-1]  # :361
    I_dr = I_dr0a[:, None, :] * np.exp(-Kb[:, None] * cols.lai)[:, :, None]  # :354-355
    return {"I_dr": I_dr, "I_df_d": dn, "I_df_u": up, "F": I_dr / mu[:, None, None] + 2 * up + 2 * dn}  # :406-418


def _coef_4s(om, R_dr0, G1, G2, mu_s):
    """Tian (2007) eq. 4 coefficients with P = 1; _solve_4s.py:188-203."""
    mu_1 = 0.5 * mu_s**2
    mu_2 = 0.5 * (1 - mu_s**2)
    alpha = 0.5 * om * (1 - mu_s) * G2
    beta = 0.5 * om * (1 - mu_s) * G1
    gamma = 0.5 * om * mu_s * G1
    eps1 = 0.25 * om * R_dr0 * mu_s
    eps2 = 0.25 * om * R_dr0 * (1 - mu_s)
    return mu_1, mu_2, alpha, beta, gamma, eps1, eps2


def solve_4s(cols, *, I_dr0, I_df0, leaf_r, leaf_t, soil_r, mu_s=0.501, method="eig", bvp_tol=1e-6,
             exact_quad=False):
    """Tian et al. (2007) four-stream; _solve_4s.py:8-293.

    y = [R2d, R1d, R1u, R2u];  y' = A y + direct * g * exp(-G x / mu0)   (:81-95)
    top (x=0):    y0 = y1 = R_top (0 for the direct problem, R_df0 for the diffuse one) (:110-116)
    bottom (x=LAI): y2 = y3 = rho/pi * (2 pi (mu1 y1 + mu2 y0) + direct mu0 pi R_dr0 e^{-G LAI/mu0}) (:128-138)

    ``method="eig"``: exact solution of the linear system (direct + diffuse problems are linear in
    their data, so they are solved once, summed).  ``method="bvp"``: the reference's numerical route
    (two ``solve_bvp`` runs per band on x = linspace(0, LAI, 50), y0 = 1) with tolerance ``bvp_tol``.
    """
    nc, nz = cols.ncol, cols.nz
    mu0 = np.cos(cols.psi)
    Gp = np.array([cols.G(c, cols.psi[c]) for c in range(nc)])  # :147
    K = Gp / mu0
    Gi = G_integrals(cols, mu_s, exact_quad=exact_quad)
    I_dr0a = _bc(I_dr0, nc)[:, 0, :]
    I_df0a = _bc(I_df0, nc)[:, 0, :]
    om = (_bc(leaf_r, nc) + _bc(leaf_t, nc))[:, 0, :]
    rho = _bc(soil_r, nc)[:, 0, :]
    nb = om.shape[-1]
    R_dr0 = I_dr0a / (PI * mu0[:, None])  # :169
    R_df0 = I_df0a / PI  # :170
    lai = cols.lai
    LAI = lai[:, 0]
    up = np.empty((nc, nz, nb))
    dn = np.empty((nc, nz, nb))

    for c in range(nc):
        G1, G2 = Gi[c]
        kap = Gp[c] / mu0[c]
        for i in range(nb):
            mu_1, mu_2, al, be, ga, e1, e2 = _coef_4s(om[c, i], R_dr0[c, i], G1, G2, mu_s)
            A = np.array([
                [(al - G2) / mu_2, be / mu_2, be / mu_2, al / mu_2],
                [be / mu_1, (ga - G1) / mu_1, ga / mu_1, be / mu_1],
                [-be / mu_1, -ga / mu_1, -(ga - G1) / mu_1, -be / mu_1],
                [-al / mu_2, -be / mu_2, -be / mu_2, -(al - G2) / mu_2],
            ])
            g = Gp[c] * np.array([e2 / mu_2, e1 / mu_1, -e1 / mu_1, -e2 / mu_2])
            x = lai[c]
            if method == "eig":
                # eigenvalues come in +-lambda pairs; for strongly scattering leaves one pair can be
                # purely imaginary (oscillatory modes) -- the reference's BVP solver integrates those
                # just the same, so keep complex arithmetic and take the real part at the end
                lam, V = np.linalg.eig(A)
                p = np.linalg.solve(A + kap * np.eye(4), -g)
                # scaled modes: growing ones anchored at x = LAI
                x0 = np.where(lam.real > 0, LAI[c], 0.0)
                mode = lambda xx: V * np.exp(lam * (xx - x0))[None, :]  # noqa: E731  (4, 4): column k = mode k
                M0, ML = mode(0.0), mode(LAI[c])
                eL = math.exp(-kap * LAI[c])
                refl = lambda Y: rho[c, i] * 2 * (mu_1 * Y[1] + mu_2 * Y[0])  # noqa: E731
                Mat = np.vstack([M0[0], M0[1], ML[2] - refl(ML), ML[3] - refl(ML)])
                src = rho[c, i] * mu0[c] * R_dr0[c, i] * eL
                rhs = np.array([
                    R_df0[c, i] - p[0], R_df0[c, i] - p[1],
                    refl(p * eL) + src - p[2] * eL, refl(p * eL) + src - p[3] * eL,
                ])
                cf = np.linalg.solve(Mat, rhs)
                E = np.exp(lam[None, :] * (x[:, None] - x0[None, :]))  # (nz, 4)
                Y = ((E * cf[None, :]) @ V.T).real + np.exp(-kap * x)[:, None] * p[None, :]  # (nz, 4)
            elif method == "bvp":
                from scipy import integrate

                xm = np.linspace(0, LAI[c], 50)
                y0 = np.ones((4, xm.size))
                Y = np.zeros((nz, 4))
                for direct, R0 in ((1, R_dr0[c, i]), (0, R_df0[c, i])):
                    fun = lambda xx, yy: A @ yy + direct * g[:, None] * np.exp(-kap * xx)[None, :]  # noqa: E731
                    top = 0.0 if direct else R0

                    def bcs(ya, yb):
                        Rr = rho[c, i] / PI * (
                            2 * PI * (mu_1 * yb[1] + mu_2 * yb[0])
                            + direct * mu0[c] * PI * R0 * math.exp(-Gp[c] * LAI[c] / mu0[c])
                        )
                        return np.array([ya[0] - top, ya[1] - top, yb[2] - Rr, yb[3] - Rr])

                    res = integrate.solve_bvp(fun, bcs, xm, y0, tol=bvp_tol)
                    Y += res.sol(x).T
            else:
                raise ValueError("method must be 'eig' or 'bvp'")
            dn[c, :, i] = 2 * PI * (mu_1 * Y[:, 1] + mu_2 * Y[:, 0])  # :246-247,281
            up[c, :, i] = 2 * PI * (mu_1 * Y[:, 2] + mu_2 * Y[:, 3])  # :248-249,280
    I_dr = I_dr0a[:, None, :] * np.exp(-K[:, None] * lai)[:, :, None]  # :284
    return {"I_dr": I_dr, "I_df_d": dn, "I_df_u": up, "F": I_dr / mu0[:, None, None] + 2 * up + 2 * dn}


SOLVERS = {"2s": solve_2s, "4s": solve_4s, "bf": solve_bf, "bl": solve_bl, "g77": solve_g77, "n79": solve_n79, "zq": solve_zq, "zq_pa": solve_zq_pa}


# ----------------------------------------------------------------------------------------------
# epilogue: layer absorption + band integral


def calc_absorption(cols, out, *, leaf_r, leaf_t):
    """Layerwise absorbed irradiance and sunlit/shaded split; model.py:573-647."""
    nc = cols.ncol
    lai = cols.lai
    K_b = cols.K_b()
    dlai = lai[:, :-1] - lai[:, 1:]  # model.py:248
    leaf_a = 1 - (_bc(leaf_r, nc) + _bc(leaf_t, nc))  # :584
    I_dr, dn, up = out["I_dr"], out["I_df_d"], out["I_df_u"]
    laim = (lai[:, :-1] + lai[:, 1:]) / 2  # :601
    f_sl = np.exp(-K_b[:, None] * laim)  # :602
    a = I_dr[:, 1:] - I_dr[:, :-1] + dn[:, 1:] - dn[:, :-1] + up[:, :-1] - up[:, 1:]  # :609
    a_dr = I_dr[:, 1:] * (1 - np.exp(-K_b[:, None] * dlai))[:, :, None] * leaf_a  # :617-621
    a_df = a - a_dr
    a_df_sl = a_df * f_sl[:, :, None]
    a_df_sh = a_df * (1 - f_sl)[:, :, None]
    return {
        "aI": a, "aI_df": a_df, "aI_dr": a_dr, "aI_sh": a_df_sh, "aI_sl": a_df_sl + a_dr,
        "aI_df_sl": a_df_sl, "aI_df_sh": a_df_sh, "laim": laim, "f_slm": f_sl,
    }  # :637-647


BAND_DEFNS_UM = {"PAR": (0.4, 0.7), "NIR": (0.7, 2.5), "UV": (0.01, 0.4), "solar": (0.3, 5.0)}  # spectra.py:22-27


def x_frac_in_bounds(xe, bounds):
    """Fractional overlap of each bin [xe_i, xe_{i+1}] with ``bounds``; spectra.py:71-126
    (known answers: reference tests/test_spectra.py:25-35)."""
    xe = np.asarray(xe, dtype=np.float64)
    x1, x2 = xe[:-1], xe[1:]
    b1, b2 = bounds
    inb = (x2 >= b1) & (x1 <= b2)
    w = np.ones_like(x1)
    left = x1 < b1
    right = (~left) & (x2 > b2)
    w = np.where(left, (x2 - b1) / (x2 - x1), w)
    w = np.where(right, (b2 - x1) / (x2 - x1), w)
    return np.where(inb, w, 0.0)


def band_sum(X, wle, band_name="PAR", bounds=None):
    """Spectral integral sum_wl w * X over the last axis; diagnostics.py:39-108 (:81)."""
    if bounds is None:
        bounds = BAND_DEFNS_UM[band_name]
    return X @ x_frac_in_bounds(wle, bounds)


def e_wl_umol(wl_um):
    """J per micromole of photons at wavelength wl_um (micrometres); spectra.py:30-39 (CODATA 2018 exact h, c, N_A, the values
    scipy.constants holds).  Known answer: reference tests/test_spectra.py:75-79."""
    wl = np.asarray(wl_um, dtype=np.float64) * 1e-6
    return 6.62607015e-34 * 299792458.0 / wl * 6.02214076e23 * 1e-6


def band_profiles(sol, absorption, wle, names=("PAR", "NIR", "solar"), *, wl=None, pfd=False):
    """Everything diagnostics.band returns for the "I..." / "F" variables of one run (diagnostics.py:39-108): for every variable
    X (..., n_wl) the stack over band groups of sum_wl w X (:81); pfd=True: of sum_wl w (X / e_wl_umol(wl)) (:19-36, :92-104)."""
    W = np.stack([x_frac_in_bounds(wle, BAND_DEFNS_UM[n]) for n in names])
    if pfd:
        wle_ = np.asarray(wle, dtype=np.float64)
        wl = 0.5 * (wle_[:-1] + wle_[1:]) if wl is None else np.asarray(wl, dtype=np.float64)
        f = 1.0 / e_wl_umol(wl)
    var = {k: sol[k] for k in ("I_dr", "I_df_d", "I_df_u", "F")}
    var["I_d"] = sol["I_dr"] + sol["I_df_d"]  # model.py:425
    var.update({k: v for k, v in absorption.items() if k.startswith("aI")})
    out = {}
    for k, X in var.items():
        Xf = X * f if pfd else X
        # (the reference names the photon-flux variants vn.replace("I", "PFD"); "F" keeps its name and is overwritten in its dataset)
        out[k.replace("I", "PFD") if pfd and k != "F" else k] = np.stack([(Xf * w).sum(axis=-1) for w in W], axis=-1)  # (..., ngroup)
    return out


# ---------------------------------------------------------------------------------------------
# Input side
def smear_tuv(x, y, bins):
    """``smear_tuv`` / ``_smear_tuv_1`` (crt1d/spectra.py:221-300): in-bin average of the piecewise-linear ``y(x)``,
    trapezoid by trapezoid in index order.  Known answers: reference tests/test_spectra.py:38-55 (crt1d.spectra itself
    cannot be imported here -- it needs xarray -- so those four cases are the pin for this function)."""
    x, y, bins = np.asarray(x, dtype=float), np.asarray(y, dtype=float), np.asarray(bins, dtype=float)
    ynew = np.zeros(bins.size - 1)
    for i in range(bins.size - 1):
        xl, xu = bins[i], bins[i + 1]
        area = 0.0
        for k in range(x.size - 1):
            if x[k + 1] < xl:  # :240
                continue
            if x[k] > xu:  # :242
                break
            a1 = max(x[k], xl)
            a2 = min(x[k + 1], xu)
            slope = (y[k + 1] - y[k]) / (x[k + 1] - x[k])
            b1 = y[k] + slope * (a1 - x[k])
            b2 = y[k] + slope * (a2 - x[k])
            area = area + (a2 - a1) * (b2 + b1) / 2  # :251
        ynew[i] = area / (xu - xl)
    return ynew


def distribute_lai_beta(h_c, LAI, n, h_min=0.5):
    """``distribute_lai_beta`` (crt1d/leaf_area.py:42-93): returns ``(lai, lad, z)``.  Pinned by tests/golden/g8_leaf_area.npz
    (outputs of the reference's own function)."""
    from scipy.stats import beta

    d = (h_c - 0.7 * h_c) / h_c
    b = 3
    a = -((b - 2) * d + 1) / (d - 1)
    frac = np.linspace(1.0, 0, n)
    z = (h_c - h_min) * (1 - beta(a, b).ppf(frac)) + h_min
    zrel = (z - h_min) / (h_c - h_min)
    lad = LAI / (h_c - h_min) * beta.pdf(zrel, b, a)
    return frac * LAI, lad, z
