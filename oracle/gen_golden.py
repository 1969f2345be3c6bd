#!/usr/bin/env python3
"""
Generate golden input/output vectors by running the REAL reference solvers
(``/root/reference/crt1d/solvers``) -- runs only in the build container, never on the GPU box.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--out tests/golden] [--only g1,g3]

``import crt1d`` itself fails here with an ordinary ``ModuleNotFoundError`` (xarray is not
installed), so an empty parent package is registered and only the NumPy/SciPy-only submodules are
imported (``crt1d.solvers``, ``crt1d.leaf_angle``, ``crt1d.leaf_area``) -- SURVEY.md section 8(c).
``crt1d.cases``/``crt1d.data`` need xarray; the default-case *inputs* are rebuilt from the
reference's packaged CSV files following ``cases.py:15-58`` and ``data/__init__.py:23-35,90-157``.

Fixtures (``.npz``, data only):

g1_default   default canopy (60 levels x 107 SPCTRAL2 bands), all 8 schemes
g2_bonan     inputs of the reference's tests/test_n79.py:13-44; n79 with '9sky' and 'quad'
g3_uniform   12 synthetic columns x 10 bands x 60 levels (equal dLAI), 7 schemes (+ 4s tol=1e-11)
g4_ragged    12 synthetic columns x 10 bands x 40 levels, NON-uniform dLAI (exposes the quirks)
g5_4s_tight  default case, 4s with solve_bvp tol 1e-11 (the reference's stock tol is 1e-6)
g7_options   mu_s in {0.501, 0.33998}; G_fn in {spherical, horizontal, vertical, ellipsoidal x in {0.5,1,2}, bonan}
g8_leaf_area reference leaf_area.distribute_lai_beta(h_c, LAI, n, h_min=...) for 8 canopies (lai, lad, z)
g9_common    reference solvers.common: tau_b_fn, tau_df_fn ('quad' and '9sky'), K_df_fn for 5 leaf-angle functions
g6_absorption reference model._calc_absorption (model.py:573-647) on the reference solvers' own profiles (default case: 2s, n79,
             zq; the 12 ragged columns of g4: 2s), reference spectra._x_frac_in_bounds weights (spectra.py:71-126) for PAR / NIR /
             UV / solar on the default and the synthetic edges, and the band sums of diagnostics.py:81 formed with those weights.
             ``crt1d.model`` / ``crt1d.spectra`` cannot be imported (top-level ``import xarray``); the two FUNCTIONS are compiled
             from their definitions in the reference files where they lie (ast) and run here -- nothing is copied.
"""

import argparse
import importlib
import os
import sys
import types
from pathlib import Path

import numpy as np
import scipy

REF = Path("/root/reference")
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def import_reference():
    pkg = types.ModuleType("crt1d")
    pkg.__path__ = [str(REF / "crt1d")]
    sys.modules["crt1d"] = pkg
    la = importlib.import_module("crt1d.leaf_angle")
    lar = importlib.import_module("crt1d.leaf_area")
    sol = importlib.import_module("crt1d.solvers")
    return la, lar, sol


META = dict(scipy=scipy.__version__, numpy=np.__version__, generator="oracle/gen_golden.py")

SCHEMES = ["2s", "4s", "bf", "bl", "g77", "n79", "zq", "zq_pa"]


def run_scheme(sol, name, p, **opts):
    """Call reference solve_<name> with exactly its declared args (as Model.run does, model.py:305-310)."""
    sd = sol.AVAILABLE_SCHEMES[name]
    args = {k: p[k] for k in sd["args"]}
    return sd["solver"](**args, **opts)


def tight_bvp(tol=1e-11):
    """Context: make the reference's solve_bvp calls use a tighter tolerance (oracle process only)."""
    import scipy.integrate as si

    class _Ctx:
        def __enter__(self):
            self.orig = si.solve_bvp

            def wrapped(fun, bc, x, y, **kw):
                kw["tol"] = tol
                kw.setdefault("max_nodes", 200000)
                res = self.orig(fun, bc, x, y, **kw)
                assert res.status == 0, res.message
                return res

            si.solve_bvp = wrapped

        def __exit__(self, *a):
            si.solve_bvp = self.orig

    return _Ctx()


def default_case_inputs(la, lar, nlayers=60):
    """cases.py:15-58 with data/__init__.py loaders restated on the packaged CSVs."""
    d = REF / "crt1d" / "data"
    res = lar.distribute_lai_beta(20.0, 4.0, nlayers)
    # ideal leaf, midpoint version (data/__init__.py:90-113)
    wl, t, r = np.loadtxt(d / "ideal-green-leaf_SPCTRAL2-wavelengths.csv", delimiter=",", skiprows=1, unpack=True)
    t[t == 0] = 1e-10
    r[r == 0] = 1e-10
    wl_m, t_m, r_m = (wl[:-1] + wl[1:]) / 2, (t[:-1] + t[1:]) / 2, (r[:-1] + r[1:]) / 2
    # SPCTRAL2 default spectrum, midpoint version (data/__init__.py:116-147)
    wl0, SI_dr0, SI_df0 = np.loadtxt(d / "SPCTRAL2_xls_default-spectrum.csv", delimiter=",", skiprows=1, unpack=True)
    dwl = np.diff(wl0)
    wl_sp = wl0[:-1] + 0.5 * dwl
    I_dr = (SI_dr0[:-1] + SI_dr0[1:]) / 2 * dwl
    I_df = (SI_df0[:-1] + SI_df0[1:]) / 2 * dwl
    assert np.allclose(wl_m, wl_sp)
    # soil (data/__init__.py:23-35)
    rs = np.ones_like(wl_m)
    rs[wl_m <= 0.7] = 0.1100
    rs[wl_m > 0.7] = 0.2250
    ok = ~(np.isnan(t_m) | np.isnan(r_m) | np.isnan(I_dr) | np.isnan(I_df))  # .dropna(dim="wl") cases.py:25
    mla = 57
    x = la.mla_to_x_approx(mla)
    G_fn = lambda psi_: la.G_ellipsoidal_approx(psi_, x)  # noqa: E731
    psi = np.deg2rad(20)
    p = dict(
        lai=res.lai, z=res.z, mla=mla, clump=1.0, orient=x, G_fn=G_fn, psi=psi,
        leaf_t=t_m[ok], leaf_r=r_m[ok], soil_r=rs[ok], I_dr0_all=I_dr[ok], I_df0_all=I_df[ok],
        wl=wl_m[ok], dwl=dwl[ok],
    )
    p["K_b_fn"] = lambda psi_: p["G_fn"](psi_) / np.cos(psi_)  # model.py:291
    return p


def save(out, name, **arrays):
    arrays = {k: np.asarray(v) for k, v in arrays.items()}
    arrays["meta"] = np.array(repr(META))
    path = out / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({path.stat().st_size/1e3:.0f} kB)")


def flatten(prefix, sol_dict):
    return {f"{prefix}__{k}": v for k, v in sol_dict.items()}


def g1(la, lar, sol, out):
    p = default_case_inputs(la, lar)
    arrays = dict(
        psi=p["psi"], lai=p["lai"], mla=p["mla"], x=p["orient"], leaf_t=p["leaf_t"], leaf_r=p["leaf_r"],
        soil_r=p["soil_r"], I_dr0_all=p["I_dr0_all"], I_df0_all=p["I_df0_all"], wl=p["wl"], dwl=p["dwl"],
        clump=p["clump"],
    )
    for s in SCHEMES:
        arrays.update(flatten(s, run_scheme(sol, s, p)))
        print("  g1", s)
    save(out, "g1_default", **arrays)
    return p


def g5(la, lar, sol, out, p):
    with tight_bvp():
        r = run_scheme(sol, "4s", p)
    save(out, "g5_4s_tight", **flatten("4s_tol1e-11", r))


def g2(la, lar, sol, out):
    lap = lar.distribute_lai_beta_bonan(20, 6, 61)
    p = dict(
        lai=lap.lai, z=lap.z, psi=30 * (np.pi / 180),
        leaf_r=np.r_[0.1, 0.45], leaf_t=np.r_[0.05, 0.25], I_dr0_all=np.r_[0.8, 0.8], I_df0_all=np.r_[0.2, 0.2],
        soil_r=np.r_[0.1, 0.2], wl=np.r_[0.55, 1.6], dwl=np.r_[0.3, 1.8], clump=1.0, G_fn=la.G_spherical,
    )
    p["K_b_fn"] = lambda psi_: p["G_fn"](psi_) / np.cos(psi_)
    arrays = {k: p[k] for k in ("lai", "z", "psi", "leaf_r", "leaf_t", "I_dr0_all", "I_df0_all", "soil_r", "wl", "dwl")}
    arrays.update(flatten("n79_9sky", run_scheme(sol, "n79", p, tau_d_method="9sky")))
    arrays.update(flatten("n79_quad", run_scheme(sol, "n79", p, tau_d_method="quad")))
    save(out, "g2_bonan", **arrays)


def synth_case(la, sol, out, name, ncol, nb, nz, uniform, schemes, with_tight_4s):
    from crt1d_amd.synth import make_columns

    cols = make_columns(ncol, nb, nz, seed=1234 if uniform else 4321, uniform_dlai=uniform)
    arrays = {k: cols[k] for k in ("psi", "lai", "mla", "g_kind", "g_param", "leaf_r", "leaf_t", "soil_r", "I_dr0", "I_df0", "wle")}
    results = {}
    for c in range(ncol):
        x = cols["g_param"][c]
        p = dict(
            psi=float(cols["psi"][c]), lai=cols["lai"][c].copy(), mla=float(cols["mla"][c]), clump=1.0,
            leaf_r=cols["leaf_r"][c], leaf_t=cols["leaf_t"][c], soil_r=cols["soil_r"][c],
            I_dr0_all=cols["I_dr0"][c], I_df0_all=cols["I_df0"][c],
            G_fn=lambda psi_, x=x: la.G_ellipsoidal_approx(psi_, x),
        )
        p["K_b_fn"] = lambda psi_, p=p: p["G_fn"](psi_) / np.cos(psi_)
        for s in schemes:
            r = run_scheme(sol, s, p)
            for k, v in r.items():
                results.setdefault(f"{s}__{k}", []).append(v)
        if with_tight_4s:
            with tight_bvp():
                r = run_scheme(sol, "4s", p)
            for k, v in r.items():
                results.setdefault(f"4s_tol1e-11__{k}", []).append(v)
        print(f"  {name} column {c+1}/{ncol}")
    arrays.update({k: np.stack(v) for k, v in results.items()})
    save(out, name, **arrays)


def g7(la, lar, sol, out):
    """Option sweep on a small case: 20 levels x 6 bands, one column per G function."""
    rng = np.random.default_rng(77)
    nz, nb = 20, 6
    lai = np.linspace(3.2, 0, nz)
    base = dict(
        psi=np.deg2rad(37.0), lai=lai, mla=45.0, clump=1.0,
        leaf_r=rng.uniform(0.05, 0.5, nb), leaf_t=rng.uniform(0.03, 0.4, nb), soil_r=rng.uniform(0.05, 0.4, nb),
        I_dr0_all=rng.uniform(0, 10, nb), I_df0_all=rng.uniform(0, 5, nb),
    )
    arrays = {k: base[k] for k in ("psi", "lai", "mla", "leaf_r", "leaf_t", "soil_r", "I_dr0_all", "I_df0_all")}
    gfuns = {
        "spherical": (la.G_spherical, 1, 0.0),
        "horizontal": (la.G_horizontal, 0, 0.0),
        "vertical": (la.G_vertical, 2, 0.0),
        "ellipsoidal_x0.5": (lambda p: la.G_ellipsoidal(p, 0.5), 3, 0.5),
        "ellipsoidal_x1": (lambda p: la.G_ellipsoidal(p, 1), 3, 1.0),
        "ellipsoidal_x2": (lambda p: la.G_ellipsoidal(p, 2.0), 3, 2.0),
        "ellipsoidal_approx_x2": (lambda p: la.G_ellipsoidal_approx(p, 2.0), 4, 2.0),
        "bonan_xl0.25": (lambda p: la.G_ellipsoidal_approx_bonan(p, 0.25), 5, 0.25),
        "bonan_xl-0.9": (lambda p: la.G_ellipsoidal_approx_bonan(p, -0.9), 5, -0.9),
    }
    names, kinds, params = [], [], []
    for gname, (gf, kind, param) in gfuns.items():
        p = dict(base, G_fn=gf)
        p["K_b_fn"] = lambda psi_, p=p: p["G_fn"](psi_) / np.cos(psi_)
        names.append(gname)
        kinds.append(kind)
        params.append(param)
        for s in ("2s", "bl", "g77", "n79", "zq"):
            if gname == "horizontal" and s in ("bl", "n79", "zq"):
                pass  # K_b_fn = 1 exactly; quad still fine
            arrays.update(flatten(f"{gname}__{s}", run_scheme(sol, s, p)))
        arrays.update(flatten(f"{gname}__n79_9sky", run_scheme(sol, "n79", p, tau_d_method="9sky")))
        for mu_s in (0.501, 0.33998):
            with tight_bvp():
                arrays.update(flatten(f"{gname}__4s_mus{mu_s}_tol1e-11", run_scheme(sol, "4s", p, mu_s=mu_s)))
        print("  g7", gname)
    arrays["g_names"] = np.array(names)
    arrays["g_kind"] = np.array(kinds, dtype=np.int32)
    arrays["g_param"] = np.array(params)
    save(out, "g7_options", **arrays)


def g8(lar, out):
    """Leaf-area profiles (leaf_area.py:42-93) -- input side, SURVEY 8(f) rank 4."""
    cases = [(20.0, 4.0, 60, 0.5), (10.0, 5.0, 20, 0.5), (35.0, 7.5, 100, 2.0), (1.2, 0.8, 5, 0.1), (20.0, 6.0, 61, 0.5),
             (8.0, 3.0, 2, 0.5), (17.3, 2.2, 33, 0.0), (50.0, 9.9, 200, 5.0)]
    d = {"h_c": np.array([c[0] for c in cases]), "LAI": np.array([c[1] for c in cases]), "n": np.array([c[2] for c in cases]),
         "h_min": np.array([c[3] for c in cases])}
    for i, (h_c, LAI, n, h_min) in enumerate(cases):
        r = lar.distribute_lai_beta(h_c, LAI, n, h_min=h_min)
        d[f"c{i}__lai"], d[f"c{i}__lad"], d[f"c{i}__z"] = r.lai, r.lad, r.z
    np.savez_compressed(out / "g8_leaf_area.npz", meta=np.array(str(META)), **d)
    print("wrote", out / "g8_leaf_area.npz")


def g9(la, out):
    """solvers/common.py:11-95 -- row a7 of the scope table."""
    com = importlib.import_module("crt1d.solvers.common")
    lai = np.r_[1e-3, 0.01, 0.1, 0.37, 1.0, 2.5, 4.0, 8.0]
    gfs = {"spherical": la.G_spherical, "horizontal": la.G_horizontal, "vertical": la.G_vertical,
           "ellipsoidal_x2": lambda p: la.G_ellipsoidal(p, 2.0), "ellipsoidal_approx_x0.96": lambda p: la.G_ellipsoidal_approx(p, 0.9632)}
    d = {"lai": lai, "psi": np.array(0.35)}
    for name, G in gfs.items():
        K = lambda p, G=G: G(p) / np.cos(p)  # noqa: E731
        d[f"{name}__tau_b"] = com.tau_b_fn(K, 0.35, lai)
        d[f"{name}__tau_d_quad"] = com.tau_df_fn(K, lai, method="quad")
        d[f"{name}__tau_d_9sky"] = com.tau_df_fn(K, lai, method="9sky")
        d[f"{name}__K_d_quad"] = np.array(com.K_df_fn(K, 4.0))
        d[f"{name}__K_d_9sky"] = np.array(com.K_df_fn(K, 4.0, method="9sky"))
        d[f"{name}__tau_d_quad_scalar"] = np.array(com.tau_df_fn(K, 2.5))
    np.savez_compressed(out / "g9_common.npz", meta=np.array(str(META)), **d)
    print("wrote", out / "g9_common.npz")


def reference_functions(path, names, extra=None):
    """Compile the named top-level function definitions / assignments of a reference source file (build container only) without
    importing the module (its top-level ``import xarray`` fails here with an ordinary ModuleNotFoundError)."""
    import ast
    import warnings as _w

    src = (REF / path).read_text()
    tree = ast.parse(src)
    keep = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name in names)
            or (isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id in names for t in n.targets))]
    assert len(keep) == len(names), [getattr(n, "name", None) for n in keep]
    ns = {"np": np, "warnings": _w}
    ns.update(extra or {})
    exec(compile(ast.Module(body=keep, type_ignores=[]), str(REF / path), "exec"), ns)
    return ns


def g6(la, lar, sol, out):
    """Row a11/a12: the reference's own `_calc_absorption` and `_x_frac_in_bounds` outputs."""
    calc = reference_functions("crt1d/model.py", ["_calc_absorption"])["_calc_absorption"]
    sp = reference_functions("crt1d/spectra.py", ["_x_frac_in_bounds", "BAND_DEFNS_UM"])
    xfrac, defs = sp["_x_frac_in_bounds"], sp["BAND_DEFNS_UM"]
    KEYS = ("aI", "aI_df", "aI_dr", "aI_sh", "aI_sl", "aI_df_sl", "aI_df_sh", "laim", "f_slm")
    p = default_case_inputs(la, lar)
    lai = p["lai"]
    p["dlai"] = lai[:-1] - lai[1:]  # model.py:248
    p["K_b"] = p["K_b_fn"](p["psi"])  # model.py:293
    wle = np.r_[p["wl"][0] - 0.5 * p["dwl"][0], p["wl"] + 0.5 * p["dwl"]]  # model.py:287
    arrays = dict(psi=p["psi"], lai=lai, x=p["orient"], leaf_r=p["leaf_r"], leaf_t=p["leaf_t"], wle=wle, K_b=p["K_b"])
    names = ("PAR", "NIR", "UV", "solar")
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # UV bounds extend below the data range: the reference warns, by design
        W = np.stack([xfrac(wle, defs[n]) for n in names])
    arrays["band_names"] = np.array(names)
    arrays["band_bounds"] = np.array([defs[n] for n in names])
    arrays["w_default"] = W
    from types import SimpleNamespace

    for s in ("2s", "n79", "zq"):
        r = run_scheme(sol, s, p)
        m = SimpleNamespace(_p=p, out={k: r[k] for k in ("I_dr", "I_df_d", "I_df_u")})
        ab = calc(m)
        # (the profiles themselves are the ones stored in g1_default.npz: same solver, same inputs, deterministic)
        for k in KEYS:
            arrays[f"{s}__{k}"] = ab[k]
        # diagnostics.py:81  ds[vn] = (da * w).sum(dim="wl")  -- formed here with NumPy on the reference's weights and arrays
        for k in ("aI", "aI_sl", "aI_sh"):
            arrays[f"{s}__{k}__bandsum"] = np.stack([(ab[k] * w).sum(axis=-1) for w in W])
        print("  g6", s)
    # ragged synthetic columns (the reference solver outputs already stored in g4) -> per-column _calc_absorption
    g4 = np.load(out / "g4_ragged.npz")
    ncol = g4["psi"].shape[0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        arrays["w_synth"] = np.stack([xfrac(g4["wle"], defs[n]) for n in names])
    res = {k: [] for k in KEYS}
    for c in range(ncol):
        x = g4["g_param"][c]
        lai_c = g4["lai"][c]
        pc = dict(lai=lai_c, dlai=lai_c[:-1] - lai_c[1:], K_b=la.G_ellipsoidal_approx(float(g4["psi"][c]), x) / np.cos(float(g4["psi"][c])),
                  leaf_r=g4["leaf_r"][c], leaf_t=g4["leaf_t"][c])
        m = SimpleNamespace(_p=pc, out={k: g4[f"2s__{k}"][c] for k in ("I_dr", "I_df_d", "I_df_u")})
        ab = calc(m)
        for k in KEYS:
            res[k].append(ab[k])
    for k in KEYS:
        arrays[f"ragged2s__{k}"] = np.stack(res[k])
    save(out, "g6_absorption", **arrays)


def g10(la, lar, sol, out):
    """Row a12 completed + the input side: what the reference's `diagnostics.band` returns for its "I..." / "F" variables
    (diagnostics.py:81 `(da * w).sum(dim="wl")`; with calc_PFD `_E_to_PFD_da` first, :19-36, :92-104), formed with the reference's own
    `_x_frac_in_bounds` and `e_wl_umol` (compiled from crt1d/spectra.py where it lies) on the reference solvers' profiles and the
    reference's `_calc_absorption`; and `smear_tuv` (spectra.py:221-300) on seeded random spectra."""
    import scipy.constants as sc
    from types import SimpleNamespace

    calc = reference_functions("crt1d/model.py", ["_calc_absorption"])["_calc_absorption"]
    sp = reference_functions("crt1d/spectra.py", ["_x_frac_in_bounds", "BAND_DEFNS_UM", "e_wl_umol", "_smear_tuv_1", "smear_tuv"],
                             extra=dict(h=sc.h, c=sc.c, N_A=sc.N_A))
    xfrac, defs, e_umol = sp["_x_frac_in_bounds"], sp["BAND_DEFNS_UM"], sp["e_wl_umol"]
    p = default_case_inputs(la, lar)
    lai = p["lai"]
    p["dlai"] = lai[:-1] - lai[1:]
    p["K_b"] = p["K_b_fn"](p["psi"])
    wl = p["wl"]
    wle = np.r_[wl[0] - 0.5 * p["dwl"][0], wl + 0.5 * p["dwl"]]  # model.py:287
    names = ("PAR", "NIR", "solar")
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        W = np.stack([xfrac(wle, defs[n]) for n in names])
    f = 1 / e_umol(wl)  # diagnostics.py:27
    arrays = dict(wl=wl, wle=wle, band_names=np.array(names), w=W, e_wl_umol=e_umol(wl), e_wl_umol_at_3um=np.array(e_umol(3)))
    for s in ("2s", "n79", "zq", "4s"):
        r = run_scheme(sol, s, p)
        prof = {k: r[k] for k in ("I_dr", "I_df_d", "I_df_u", "F")}
        prof["I_d"] = r["I_dr"] + r["I_df_d"]  # model.py:425
        m = SimpleNamespace(_p=p, out={k: r[k] for k in ("I_dr", "I_df_d", "I_df_u")})
        prof.update({k: v for k, v in calc(m).items() if k.startswith("aI")})
        for k, da in prof.items():
            arrays[f"{s}__{k}__band"] = np.stack([(da * w).sum(axis=-1) for w in W])                 # :81
            # :93-97 (da_pfd = da * f, :31).  (In the dataset the reference names it vn.replace("I", "PFD") -- which for "F" is "F"
            # again: with calc_PFD its F is overwritten by the photon-flux version.  Stored here under separate keys.)
            arrays[f"{s}__{k}__band_pfd"] = np.stack([((da * f) * w).sum(axis=-1) for w in W])
        print("  g10", s)
    rng = np.random.default_rng(77)
    x = np.sort(rng.uniform(0.28, 2.7, 400))
    y = rng.uniform(0.0, 2.0, (5, 400)) * np.exp(-((x - 0.9) ** 2))[None, :]
    bins = np.r_[0.25, np.sort(rng.uniform(0.3, 2.6, 40)), 2.9]  # first / last bin reach beyond the data
    arrays.update(smear_x=x, smear_y=y, smear_bins=bins, smear_out=np.stack([sp["smear_tuv"](x, yy, bins) for yy in y]))
    save(out, "g10_band_profiles", **arrays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(REPO / "tests" / "golden"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    out = Path(a.out)
    out.mkdir(parents=True, exist_ok=True)
    only = set(a.only.split(",")) if a.only else None
    want = lambda k: only is None or k in only  # noqa: E731
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import warnings

    warnings.simplefilter("ignore")
    la, lar, sol = import_reference()
    p = None
    if want("g1") or want("g5"):
        p = default_case_inputs(la, lar)
    if want("g1"):
        g1(la, lar, sol, out)
    if want("default-data"):
        # spectral inputs of the default case as a small packaged data file for crt1d_amd.cases
        pd_ = default_case_inputs(la, lar)
        path = REPO / "crt1d_amd" / "data" / "default_spectra.npz"
        np.savez_compressed(
            path, **{k: np.asarray(pd_[k]) for k in ("wl", "dwl", "leaf_t", "leaf_r", "soil_r", "I_dr0_all", "I_df0_all")},
            meta=np.array("default spectra of crt1d (ideal green leaf + SPCTRAL2 default spectrum + 2-value soil), NaN bands "
                          "dropped; derived by oracle/gen_golden.py from the reference's packaged CSV files"),
        )
        print("wrote", path)
    if want("g5"):
        g5(la, lar, sol, out, p)
    if want("g2"):
        g2(la, lar, sol, out)
    if want("g3"):
        synth_case(la, sol, out, "g3_uniform", 12, 10, 60, True, ["2s", "4s", "bf", "bl", "g77", "n79", "zq", "zq_pa"], True)
    if want("g4"):
        synth_case(la, sol, out, "g4_ragged", 12, 10, 40, False, ["2s", "4s", "bf", "bl", "g77", "n79", "zq", "zq_pa"], True)
    if want("g6"):
        g6(la, lar, sol, out)
    if want("g7"):
        g7(la, lar, sol, out)
    if want("g8"):
        g8(lar, out)
    if want("g9"):
        g9(la, out)
    if want("g10"):
        g10(la, lar, sol, out)


if __name__ == "__main__":
    main()
