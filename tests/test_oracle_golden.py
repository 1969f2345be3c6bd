"""CPU: pin the oracle (NumPy restatement) to golden vectors produced by the REAL reference solvers.

Two modes per quadrature-using scheme:
  * oracle with the reference's own scipy.integrate.quad calls (exact_quad=True): agrees to rounding (<= 1e-11)
    -> the restated algorithm IS the reference's;
  * oracle with its fixed Gauss-Legendre rule (what it is used with everywhere else): differs from the reference
    only by the reference's QUADPACK error.
"""
import numpy as np
import pytest

from conftest import load_golden, rel_profile_err


def _cols(O, g, c=None):
    if "g_kind" in g.files and g["g_kind"].ndim == 1 and "x" not in g.files:
        sl = slice(None) if c is None else slice(c, c + 1)
        return O.Columns(g["psi"][sl], g["lai"][sl], mla=g["mla"][sl], g_kind=g["g_kind"][sl], g_param=g["g_param"][sl])
    return O.Columns([float(g["psi"])], g["lai"][None], mla=[float(g["mla"])], g_kind=[4], g_param=[float(g["x"])])


def _kw(g, scheme, sl=slice(None)):
    if "I_dr0" in g.files:
        kw = dict(I_dr0=g["I_dr0"][sl], I_df0=g["I_df0"][sl], leaf_r=g["leaf_r"][sl], leaf_t=g["leaf_t"][sl], soil_r=g["soil_r"][sl])
    else:
        kw = dict(I_dr0=g["I_dr0_all"], I_df0=g["I_df0_all"], leaf_r=g["leaf_r"], leaf_t=g["leaf_t"], soil_r=g["soil_r"])
    if scheme == "bl":
        kw.pop("soil_r")
    return kw


@pytest.mark.parametrize("scheme,tol", [("2s", 1e-11), ("g77", 1e-12), ("bf", 1e-12), ("bl", 1e-6), ("n79", 1e-8), ("zq", 1e-8), ("zq_pa", 1e-8)])
def test_default_case(oracle, scheme, tol):
    g = load_golden("g1_default")
    res = oracle.SOLVERS[scheme](_cols(oracle, g), **_kw(g, scheme))
    for k, v in res.items():
        if k == "rho_c":
            assert abs(v[0] - g["bf__rho_c"]) < 1e-15
            continue
        assert rel_profile_err(v[0], g[f"{scheme}__{k}"]) <= tol, k


@pytest.mark.parametrize("scheme", ["2s", "bl", "n79", "zq", "zq_pa"])
def test_default_case_exact_quad(oracle, scheme):
    """With the reference's own QUADPACK calls the restatement agrees with the reference to rounding."""
    g = load_golden("g1_default")
    res = oracle.SOLVERS[scheme](_cols(oracle, g), **_kw(g, scheme), exact_quad=True)
    for k, v in res.items():
        assert rel_profile_err(v[0], g[f"{scheme}__{k}"]) <= 1e-11, k


def test_default_case_spot_values(oracle):
    """Spot values recorded independently in SURVEY.md section 8(c)."""
    g = load_golden("g1_default")
    assert g["2s__I_df_d"][0, 10] == pytest.approx(0.10209515394426368, rel=1e-13)
    assert g["n79__I_df_u"][-1, 10] == pytest.approx(0.25550281617320236, rel=1e-13)
    assert g["zq__F"][30, 50] == pytest.approx(10.33708657748977, rel=1e-13)
    assert g["4s__I_df_d"][0, 10] == pytest.approx(0.1921645387673509, rel=1e-12)
    res = oracle.solve_2s(_cols(oracle, g), **_kw(g, "2s"))
    assert res["F"][0, 30, 50] == pytest.approx(10.62202610797509, rel=1e-12)


def test_4s_exact_vs_tight_reference(oracle):
    g, g5 = load_golden("g1_default"), load_golden("g5_4s_tight")
    res = oracle.solve_4s(_cols(oracle, g), **_kw(g, "4s"))
    for k, v in res.items():
        assert rel_profile_err(v[0], g5[f"4s_tol1e-11__{k}"]) <= 1e-9, k  # vs solve_bvp tol=1e-11
        assert rel_profile_err(v[0], g[f"4s__{k}"]) <= 2e-4, k  # stock reference is only ~1e-4 accurate


def test_bonan_n79(oracle):
    """Inputs of the reference's tests/test_n79.py:13-44; '9sky' needs no adaptive quadrature -> rounding-level."""
    g = load_golden("g2_bonan")
    cols = oracle.Columns([float(g["psi"])], g["lai"][None], g_kind=[1], g_param=[0.0])
    kw = dict(I_dr0=g["I_dr0_all"], I_df0=g["I_df0_all"], leaf_r=g["leaf_r"], leaf_t=g["leaf_t"], soil_r=g["soil_r"])
    for method, tol in (("9sky", 1e-13), ("quad", 1e-8)):
        res = oracle.solve_n79(cols, **kw, tau_d_method=method)
        for k, v in res.items():
            assert rel_profile_err(v[0], g[f"n79_{method}__{k}"]) <= tol, (method, k)
    with pytest.raises(ValueError):
        oracle.solve_n79(cols, **kw, tau_d_method="simpson")


@pytest.mark.parametrize("name", ["g3_uniform", "g4_ragged"])
@pytest.mark.parametrize("scheme", ["2s", "4s", "bf", "bl", "g77", "n79", "zq", "zq_pa"])
def test_synthetic(oracle, name, scheme):
    g = load_golden(name)
    res = oracle.SOLVERS[scheme](_cols(oracle, g), **_kw(g, scheme))
    pre = "4s_tol1e-11" if scheme == "4s" else scheme
    tol = {"2s": 1e-10, "g77": 1e-12, "bf": 1e-12, "4s": 1e-9, "bl": 1e-6, "zq": 1e-6, "n79": 1e-6, "zq_pa": 1e-6}[scheme]
    for k, v in res.items():
        if k == "rho_c":
            continue
        t = 2e-5 if (scheme == "n79" and k.startswith("aI") and name == "g4_ragged") else tol  # QUADPACK error / small dlai
        assert rel_profile_err(v, g[f"{pre}__{k}"]) <= t, k


def test_ragged_n79_zq_exact_quad(oracle):
    """Non-uniform dLAI exposes the reference's index quirks (n79 first downward row uses index 1; zq uses one mean dLAI):
    with the same QUADPACK calls the restatement reproduces them to rounding."""
    g = load_golden("g4_ragged")
    sl = slice(0, 4)
    cols = oracle.Columns(g["psi"][sl], g["lai"][sl], mla=g["mla"][sl], g_kind=g["g_kind"][sl], g_param=g["g_param"][sl])
    for scheme in ("n79", "zq"):
        res = oracle.SOLVERS[scheme](cols, **_kw(g, scheme, sl), exact_quad=True)
        for k, v in res.items():
            assert rel_profile_err(v, g[f"{scheme}__{k}"][sl]) <= 1e-11, (scheme, k)


def test_g_kinds_and_options(oracle):
    g = load_golden("g7_options")
    kw = dict(I_dr0=g["I_dr0_all"], I_df0=g["I_df0_all"], leaf_r=g["leaf_r"], leaf_t=g["leaf_t"], soil_r=g["soil_r"])
    for i, gname in enumerate(g["g_names"]):
        cols = oracle.Columns([float(g["psi"])], g["lai"][None], mla=[float(g["mla"])], g_kind=[int(g["g_kind"][i])],
                              g_param=[float(g["g_param"][i])])
        for scheme, opts, pre, tol in [
            ("2s", {}, "2s", 1e-9), ("g77", {}, "g77", 1e-12), ("bl", {}, "bl", 1e-6), ("n79", {}, "n79", 1e-6),
            ("n79", {"tau_d_method": "9sky"}, "n79_9sky", 1e-12), ("zq", {}, "zq", 1e-6),
            ("4s", {"mu_s": 0.501}, "4s_mus0.501_tol1e-11", 1e-9), ("4s", {"mu_s": 0.33998}, "4s_mus0.33998_tol1e-11", 1e-9),
        ]:
            k2 = dict(kw)
            if scheme == "bl":
                k2.pop("soil_r")
            res = oracle.SOLVERS[scheme](cols, **k2, **opts)
            for k, v in res.items():
                assert rel_profile_err(v[0], g[f"{gname}__{pre}__{k}"]) <= tol, (gname, scheme, k)


def test_invariants(oracle):
    """Known-answer invariants observed on the reference."""
    g = load_golden("g1_default")
    mu = np.cos(float(g["psi"]))
    for s in ("2s", "4s", "n79", "zq", "bl", "g77", "bf"):
        F = g[f"{s}__I_dr"] / mu + 2 * g[f"{s}__I_df_u"] + 2 * g[f"{s}__I_df_d"]
        assert np.max(np.abs(F - g[f"{s}__F"])) < 1e-13
        assert np.array_equal(g[f"{s}__I_dr"], g["2s__I_dr"])  # I_dr = I_dr0 exp(-K_b lai) for every scheme
    assert np.max(np.abs(g["2s__I_df_d"][-1] - g["I_df0_all"])) < 1e-12  # top BC
    assert np.max(np.abs(g["n79__I_df_u"][0] - g["soil_r"] * (g["n79__I_df_d"][0] + g["n79__I_dr"][0]))) < 1e-14


def test_x_frac_in_bounds_known_answers(oracle):
    """The reference's own known answers, tests/test_spectra.py:25-35."""
    np.testing.assert_allclose(oracle.x_frac_in_bounds(np.r_[0, 1, 2, 3], (0, 3)), [1, 1, 1])
    np.testing.assert_allclose(oracle.x_frac_in_bounds(np.r_[0, 1, 2, 3], (0.5, 2.2)), [0.5, 1, 0.2])
    np.testing.assert_allclose(oracle.x_frac_in_bounds(np.r_[0, 1, 2, 3], (0.5, 2.0)), [0.5, 1, 0])


def test_calc_absorption_consistency(oracle):
    """model.py:573-647 restated: a_sl + a_sh == a (the reference's own assertion, :635) and energy closure."""
    g = load_golden("g1_default")
    cols = _cols(oracle, g)
    out = {k: g[f"2s__{k}"][None] for k in ("I_dr", "I_df_d", "I_df_u")}
    ab = oracle.calc_absorption(cols, out, leaf_r=g["leaf_r"], leaf_t=g["leaf_t"])
    assert np.allclose(ab["aI_sl"] + ab["aI_sh"], ab["aI"])
    # sum of layer absorption == net flux in at top minus net flux out at the bottom
    top = out["I_dr"][0, -1] + out["I_df_d"][0, -1] - out["I_df_u"][0, -1]
    bot = out["I_dr"][0, 0] + out["I_df_d"][0, 0] - out["I_df_u"][0, 0]
    assert np.max(np.abs(ab["aI"][0].sum(axis=0) - (top - bot))) < 1e-12


@pytest.mark.parametrize(
    "x,bins,expected",
    [
        (np.r_[0, 1, 2, 3, 4], (0, 2, 4), [1, 3]),  # equal bin edges in x
        (np.r_[0, 1, 2, 3, 4], (0, 1, 4), [0.5, 2.5]),  # unequal bin edges in x
        (np.r_[0, 1, 2, 3, 4], (0, 1, 6), [0.5, 7.5 / (6 - 1)]),  # bin edge beyond x
        (np.r_[0, 1], (2, 3), [0]),  # single bin fully outside
    ],
)
def test_smear_tuv_known_answers(oracle, x, bins, expected):
    """The reference's own known answers (tests/test_spectra.py:38-55; y = x is the spectrum)."""
    np.testing.assert_allclose(oracle.smear_tuv(x, x, bins), expected)


def test_smear_tuv_conserves_the_integral(oracle):
    """Docstring property of smear_tuv (spectra.py:263-266): sum(ynew * dx) == trapezoidal integral over the bins' range."""
    rng = np.random.default_rng(3)
    x = np.cumsum(rng.uniform(0.5, 2.0, 400)) * 1e-3 + 0.3
    y = rng.uniform(0, 1, 400)
    bins = np.linspace(x[0], x[-1], 38)
    ynew = oracle.smear_tuv(x, y, bins)
    np.testing.assert_allclose((ynew * np.diff(bins)).sum(), np.trapezoid(y, x), rtol=1e-12)


def test_distribute_lai_beta_vs_reference(oracle):
    g = load_golden("g8_leaf_area")
    for i in range(len(g["h_c"])):
        lai, lad, z = oracle.distribute_lai_beta(float(g["h_c"][i]), float(g["LAI"][i]), int(g["n"][i]), float(g["h_min"][i]))
        assert np.array_equal(lai, g[f"c{i}__lai"])
        np.testing.assert_allclose(z, g[f"c{i}__z"], rtol=1e-14)
        np.testing.assert_allclose(lad, g[f"c{i}__lad"], rtol=1e-12, atol=1e-300)


def test_tau_d_vs_reference_common(oracle):
    """oracle.tau_d (fixed-node rule / 9sky) against the reference's tau_df_fn (tests/golden/g9_common.npz)."""
    g = load_golden("g9_common")
    kinds = {"spherical": (1, 0.0), "horizontal": (0, 0.0), "vertical": (2, 0.0), "ellipsoidal_x2": (3, 2.0), "ellipsoidal_approx_x0.96": (4, 0.9632)}
    for name, (kind, param) in kinds.items():
        cols = oracle.Columns([0.35], np.array([[1.0, 0.0]]), g_kind=[kind], g_param=[param])
        np.testing.assert_allclose(oracle.tau_d(cols, g["lai"][None], method="9sky")[0], g[f"{name}__tau_d_9sky"], rtol=1e-13)
        # 'quad': the reference's QUADPACK result carries its own error (up to ~3e-8, vertical leaves at small LAI)
        np.testing.assert_allclose(oracle.tau_d(cols, g["lai"][None], method="quad")[0], g[f"{name}__tau_d_quad"], rtol=2e-7)


# ---- G6: the reference's own `_calc_absorption` (model.py:573-647) and `_x_frac_in_bounds` (spectra.py:71-126) outputs,
# produced by oracle/gen_golden.py from the reference's function definitions (SURVEY.md section 8(c) fixture G6)
ABS_KEYS = ("aI", "aI_df", "aI_dr", "aI_sh", "aI_sl", "aI_df_sl", "aI_df_sh", "laim", "f_slm")


@pytest.mark.parametrize("scheme", ["2s", "n79", "zq"])
def test_g6_calc_absorption_default_case(oracle, scheme):
    g1, g6 = load_golden("g1_default"), load_golden("g6_absorption")
    cols = _cols(oracle, g1)
    out = {k: g1[f"{scheme}__{k}"][None] for k in ("I_dr", "I_df_d", "I_df_u")}
    ab = oracle.calc_absorption(cols, out, leaf_r=g1["leaf_r"], leaf_t=g1["leaf_t"])
    scale = np.abs(g6[f"{scheme}__aI"]).max()
    for k in ABS_KEYS:
        ref = g6[f"{scheme}__{k}"]
        assert np.max(np.abs(ab[k][0] - ref)) <= 1e-13 * max(scale, np.abs(ref).max()), k
    # band weights and the band sums of diagnostics.py:81
    names = [str(n) for n in g6["band_names"]]
    for gi, n in enumerate(names):
        w = oracle.x_frac_in_bounds(g6["wle"], tuple(g6["band_bounds"][gi]))
        np.testing.assert_array_equal(w, g6["w_default"][gi])  # bit-exact: same IEEE operations
        for k in ("aI", "aI_sl", "aI_sh"):
            ref = g6[f"{scheme}__{k}__bandsum"][gi]
            got = ab[k][0] @ w
            assert np.max(np.abs(got - ref)) <= 1e-12 * max(np.abs(ref).max(), 1e-300), (n, k)


def test_g6_calc_absorption_ragged_columns(oracle):
    g4, g6 = load_golden("g4_ragged"), load_golden("g6_absorption")
    cols = _cols(oracle, g4)
    out = {k: g4[f"2s__{k}"] for k in ("I_dr", "I_df_d", "I_df_u")}
    ab = oracle.calc_absorption(cols, out, leaf_r=g4["leaf_r"], leaf_t=g4["leaf_t"])
    for k in ABS_KEYS:
        ref = g6[f"ragged2s__{k}"]
        assert np.max(np.abs(ab[k] - ref)) <= 1e-13 * max(1.0, np.abs(ref).max()), k
    for gi in range(4):
        np.testing.assert_array_equal(oracle.x_frac_in_bounds(g4["wle"], tuple(g6["band_bounds"][gi])), g6["w_synth"][gi])


def test_ref_shaped_2s_equals_oracle(oracle):
    """oracle/ref_shaped.solve_2s_loop (per-band Python loop: the CPU baseline's 'reference-shaped' leg) vs the vectorised
    restatement and vs the reference's golden output."""
    from oracle import ref_shaped

    g = load_golden("g1_default")
    cols = _cols(oracle, g)
    res = ref_shaped.solve_2s_loop(psi=float(g["psi"]), lai=g["lai"], mla=float(g["mla"]), K_b=cols.K_b()[0], mu_bar=oracle.mu_bar(cols)[0],
                                   I_dr0=g["I_dr0_all"], I_df0=g["I_df0_all"], leaf_r=g["leaf_r"], leaf_t=g["leaf_t"], soil_r=g["soil_r"])
    vec = oracle.solve_2s(cols, **_kw(g, "2s"))
    for k, v in res.items():
        assert rel_profile_err(v, vec[k][0]) <= 1e-13, k
        assert rel_profile_err(v, g[f"2s__{k}"]) <= 1e-11, k


def test_band_profiles_and_pfd_vs_reference_g10(oracle):
    """Row a12 complete: the oracle's restatement of diagnostics.band (every "I..." / "F" variable, W m-2 and photon-flux variants)
    against values formed with the REFERENCE's own _x_frac_in_bounds / e_wl_umol on the reference's profiles (g10, oracle/gen_golden.py)."""
    g1, g10 = load_golden("g1_default"), load_golden("g10_band_profiles")
    O = oracle
    wl, wle = g10["wl"], g10["wle"]
    assert np.array_equal(O.e_wl_umol(wl), g10["e_wl_umol"])  # bit for bit: same constants, same operation order
    # the reference's own known answer (tests/test_spectra.py:75-79)
    assert abs(float(O.e_wl_umol(3)) / (6.621e-20 * 6.02214076e23 / 1e6) - 1) < 1e-4
    names = [str(n) for n in g10["band_names"]]
    cols = O.Columns(np.atleast_1d(g1["psi"]), g1["lai"][None, :], mla=np.atleast_1d(g1["mla"]), g_kind=np.array([4], dtype=np.int32),
                     g_param=np.atleast_1d(g1["x"]))
    for s in ("2s", "n79", "zq"):
        sol = {k: g1[f"{s}__{k}"][None] for k in ("I_dr", "I_df_d", "I_df_u", "F")}
        ab = O.calc_absorption(cols, sol, leaf_r=g1["leaf_r"][None], leaf_t=g1["leaf_t"][None])
        for pfd in (False, True):
            got = O.band_profiles(sol, ab, wle, names, wl=wl, pfd=pfd)
            for k, v in got.items():
                ref = g10[f"{s}__{k.replace('PFD', 'I')}__band" + ("_pfd" if pfd else "")]  # (ngroup, nlev)
                err = np.abs(v[0].T - ref).max() / np.abs(ref).max()
                assert err <= 1e-13, (s, k, pfd, err)


def test_smear_tuv_vs_reference_g10(oracle):
    """oracle.smear_tuv against the reference's smear_tuv (spectra.py:221-300, compiled where it lies by oracle/gen_golden.py) on
    seeded random spectra with bins reaching beyond the data: identical arithmetic, identical bits."""
    g10 = load_golden("g10_band_profiles")
    for y, ref in zip(g10["smear_y"], g10["smear_out"]):
        assert np.array_equal(oracle.smear_tuv(g10["smear_x"], y, g10["smear_bins"]), ref)
