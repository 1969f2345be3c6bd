"""CPU: host-side mirror of the reference interface (registry, contract lists, leaf angles, Model bookkeeping)."""
import warnings

import numpy as np
import pytest


def test_registry_matches_reference_introspection():
    """args / options / names exactly as the reference's registry reports them for its own modules
    (probed with the real reference; recorded in SURVEY.md section 8(a9) and oracle/gen_golden.py)."""
    from crt1d_amd import solvers

    common = ["psi", "I_dr0_all", "I_df0_all", "lai", "leaf_t", "leaf_r"]
    expect = {
        "2s": (common + ["soil_r", "K_b_fn", "G_fn", "mla"], [], "2s", "Dickinson–Sellers two-stream"),
        "4s": (common + ["soil_r", "K_b_fn", "G_fn"], ["mu_s"], "4s", "Tian et al. four-stream"),
        "bf": (common + ["soil_r", "K_b_fn"], [], "BF", "Bodin & Franklin improved Goudriaan"),
        "bl": (common + ["K_b_fn"], [], "B–L", "Beer–Lambert"),
        "g77": (common + ["soil_r", "K_b_fn"], [], "G77", "Goudriaan (1977)"),
        "n79": (common + ["soil_r", "K_b_fn"], ["tau_d_method"], "N79", "Norman (1979)"),
        "zq": (common + ["soil_r", "K_b_fn", "G_fn"], [], "ZQ", "Zhao & Qualls multi-scattering"),
        "zq_pa": (["psi", "I_dr0_all", "I_df0_all", "lai", "clump", "leaf_t", "leaf_r", "soil_r", "K_b_fn"], [], "ZQ-pA",
                  "Zhao & Qualls multi-scattering (pyAPES)"),
    }
    assert sorted(solvers.AVAILABLE_SCHEMES) == sorted(expect)
    for k, (args, opts, sn, ln) in expect.items():
        d = solvers.AVAILABLE_SCHEMES[k]
        assert d["args"] == args and d["options"] == opts and d["short_name"] == sn and d["long_name"] == ln
        assert d["name"] == k and d["module_name"] == f"_solve_{k}" and d["solver"].__name__ == f"solve_{k}"
        assert getattr(solvers, f"solve_{k}") is d["solver"]
    assert solvers.RET_KEYS_ALL_SCHEMES == ["I_dr", "I_df_d", "I_df_u", "F"]
    assert solvers.CANOPY_RAD_STATE_INPUT_KEYS == [
        "psi", "I_dr0_all", "I_df0_all", "lai", "clump", "leaf_t", "leaf_r", "soil_r", "K_b", "K_b_fn", "G", "G_fn", "mla"]
    import inspect

    sig = inspect.signature(solvers.solve_4s)
    assert sig.parameters["mu_s"].default == 0.501 and all(p.kind is p.KEYWORD_ONLY for p in sig.parameters.values())
    assert inspect.signature(solvers.solve_n79).parameters["tau_d_method"].default == "quad"


def test_leaf_angle_closed_forms(oracle):
    from crt1d_amd import leaf_angle as la

    psi = np.linspace(0, np.pi / 2 - 1e-6, 23)
    for kind, param in [(0, 0), (1, 0), (2, 0), (3, 0.5), (3, 1.0), (3, 2.0), (4, 0.9632), (5, 0.25), (5, -0.9)]:
        np.testing.assert_allclose(la.eval_G(kind, param, psi), oracle._G_closed_form(kind, param, psi), rtol=1e-15, atol=1e-16)
    # tan-free ellipsoidal form == the textbook sqrt(x^2 + tan^2)/p2 * cos form
    x = 0.9632
    ref = np.sqrt(x**2 + np.tan(psi) ** 2) / (x + 1.774 * (x + 1.182) ** -0.733) * np.cos(psi)
    np.testing.assert_allclose(la.G_ellipsoidal_approx(psi, x), ref, rtol=1e-13)
    assert la.G_spherical(0.3) == 0.5 and la.describe_G(la.G_spherical) == (la.G_SPHERICAL, 0.0)
    assert la.describe_G(lambda p: 0.5) is None
    assert abs(la.mla_to_x_approx(57) - 0.9632) < 1e-4  # default case, cases.py:29
    np.testing.assert_allclose(la.x_to_mla_approx(la.mla_to_x_approx(41.0)), 41.0)
    with pytest.raises(AssertionError):
        la.mla_to_x_approx(500.0)


def test_table_sampling_of_arbitrary_callables():
    from crt1d_amd import _lib
    from crt1d_amd.solvers import common

    G = lambda p: 0.3 + 0.2 * np.cos(p)  # noqa: E731
    Kb = lambda p: G(p) / np.cos(p)  # noqa: E731
    d = common._describe(0.4, Kb, None, 0.501)
    nodes = _lib.quad_nodes(0.501)
    assert d["g_kind"] == 6 and d["g_table"].shape == (_lib.NQ,)
    np.testing.assert_allclose(d["g_table"], G(nodes), rtol=1e-12)
    assert d["g_at_psi"] == pytest.approx(G(0.4), rel=1e-15)
    # scalar-only callables and constant-returning ones
    import math

    d2 = common._describe(0.4, lambda p: (0.3 + 0.2 * math.cos(p)) / math.cos(p), None, 0.501)
    np.testing.assert_allclose(d2["g_table"], d["g_table"], rtol=1e-12)
    d3 = common._describe(0.4, lambda p: 0.5 / np.cos(p), lambda p: 0.5, 0.501)
    np.testing.assert_allclose(d3["g_table"], 0.5)
    # tagged K_b_fn -> closed form on device, no table
    from crt1d_amd.leaf_angle import GFunction

    d4 = common._describe(0.4, common.KbFunction(GFunction(4, 1.3)), None, 0.501)
    assert d4["g_kind"] == 4 and d4["g_param"] == 1.3 and d4["g_table"] is None
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        common._describe(0.4, lambda p: 0.7 / np.cos(p), lambda p: 0.5 + 0 * p, 0.501)
        assert any("disagree" in str(x.message) for x in w)


def test_model_bookkeeping_without_gpu():
    """Everything in Model except run()/calc_absorption() is host logic (model.py:68-294)."""
    from crt1d_amd.model import Model

    m = Model("n79", nlayers=60)
    p = m.copy_p()
    assert m.nlev == 60 and m.nwl == 107 and repr(m) == "Model(scheme='n79', psi=0.3491)"
    assert p["lai"][0] == 4.0 and p["lai"][-1] == 0 and p["lai_tot"] == 4.0
    np.testing.assert_allclose(p["dlai"], 4.0 / 59)
    assert p["wle"].size == 108 and p["mu"] == np.cos(p["psi"])
    assert p["K_b"] == pytest.approx(0.520803, abs=5e-7) and p["G"] == pytest.approx(0.489395, abs=5e-7)
    assert p["K_b_fn"](0.3) == pytest.approx(p["G_fn"](0.3) / np.cos(0.3))
    # unknown scheme -> message + fallback to 2s (model.py:157-168)
    assert Model("nope").scheme["name"] == "2s"
    # bad update -> warning + revert (model.py:184-201)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.update_p(lai=p["lai"][::-1])
        assert any("Reverting" in str(x.message) for x in w)
    assert m.copy_p()["lai"][0] == 4.0
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.update_p(not_a_param=1)
        assert any("not intended as an input" in str(x.message) for x in w)
    m.update_p(psi=0.5)
    assert m.copy_p()["mu"] == np.cos(0.5)
    with pytest.raises(Exception, match="Must run the model first"):
        m.calc_absorption()


def test_default_case_matches_reference_inputs():
    from conftest import load_golden
    from crt1d_amd.cases import load_default_case

    g = load_golden("g1_default")
    p = load_default_case(60)
    for k in ("lai", "leaf_t", "leaf_r", "soil_r", "I_dr0_all", "I_df0_all", "wl", "dwl"):
        np.testing.assert_array_equal(p[k], g[k])
    assert p["psi"] == float(g["psi"]) and p["orient"] == pytest.approx(float(g["x"]), rel=1e-15)
    assert np.all(np.diff(p["z"]) > 0)


def test_band_weights_and_sharding():
    from crt1d_amd import spectra
    from crt1d_amd.dist import block_range

    np.testing.assert_allclose(spectra.x_frac_in_bounds(np.r_[0, 1, 2, 3], (0.5, 2.2)), [0.5, 1, 0.2])  # ref tests/test_spectra.py:29
    np.testing.assert_allclose(spectra.x_frac_in_bounds(np.r_[0, 1, 2, 3], (0.5, 2.0)), [0.5, 1, 0])
    w = spectra.band_weights(np.linspace(0.3, 2.6, 301))
    assert w.shape == (3, 300) and np.all(w[2] == 1.0) and np.all(w[0] + w[1] <= 1 + 1e-12)
    # 300 bands on 8 ranks -> 38,38,38,38,37,37,37,37
    sizes = [block_range(300, r, 8)[1] - block_range(300, r, 8)[0] for r in range(8)]
    assert sizes == [38, 38, 38, 38, 37, 37, 37, 37]
    assert [block_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_synth_generator():
    from crt1d_amd import synth

    d = synth.make_columns(50, 7, 11, uniform_dlai=False)
    assert d["lai"].shape == (50, 11) and np.all(d["lai"][:, -1] == 0) and np.all(np.diff(d["lai"], axis=1) < 0)
    assert np.all(d["leaf_r"] + d["leaf_t"] <= 0.95 + 1e-12) and d["g_kind"].dtype == np.int32
    d2 = synth.make_columns(50, 7, 11, uniform_dlai=False)
    assert all(np.array_equal(d[k], d2[k]) for k in d)  # seeded
    assert synth.make_columns(3, 5, 4, per_column_optics=False)["leaf_r"].shape == (1, 5)


def test_no_silent_cpu_path():
    """Without a GPU the solvers must refuse loudly, never fall back."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from crt1d_amd.model import Model

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Model("2s").run()


def test_band_partition_deals_pairs():
    """300 bands on 8 ranks -> 38 x 6 + 36 x 2 (every shard even: the fused 16-byte flush applies on every rank), contiguous and
    complete; odd totals and tiny spectra fall back to single bands."""
    from crt1d_amd.dist import band_block_range, block_range

    r = [band_block_range(300, k, 8) for k in range(8)]
    assert [hi - lo for lo, hi in r] == [38] * 6 + [36] * 2
    assert r[0][0] == 0 and r[-1][1] == 300 and all(r[i][1] == r[i + 1][0] for i in range(7))
    assert [band_block_range(9, k, 2) for k in range(2)] == [(0, 5), (5, 9)]
    assert [band_block_range(10, k, 8)[1] - band_block_range(10, k, 8)[0] for k in range(8)] == [2, 2, 1, 1, 1, 1, 1, 1]
    assert [block_range(7, k, 3) for k in range(3)] == [(0, 3), (3, 5), (5, 7)]


def test_canopy_derive_is_pure_and_validates():
    """crt1d_amd.canopy.derive (the validation / derivation of model.py:222-294 as a pure function): does not touch its argument, derives
    the documented entries, and refuses the cases the reference asserts on."""
    from crt1d_amd import canopy
    from crt1d_amd.cases import load_default_case

    p = load_default_case(nlayers=12)
    before = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in p.items()}
    d = canopy.derive(p)
    assert set(p) == set(before) and all(np.array_equal(p[k], before[k]) if isinstance(before[k], np.ndarray) else p[k] is before[k] or p[k] == before[k] for k in before)
    assert set(d) == {"lai_tot", "lai_eff", "dlai", "dlai_eff", "zm", "dz", "mu", "wle", "K_b_fn", "G", "K_b"}
    np.testing.assert_allclose(d["dlai"].sum(), p["lai"][0])
    np.testing.assert_allclose(d["zm"], 0.5 * (p["z"][:-1] + p["z"][1:]))
    np.testing.assert_allclose(np.diff(d["wle"]), p["dwl"])
    assert d["K_b"] == pytest.approx(d["G"] / np.cos(p["psi"]))
    assert canopy.sizes(p) == (12, p["wl"].size)
    for bad in (dict(lai=p["lai"][::-1]), dict(z=p["z"][::-1]), dict(lai=p["lai"] + 0.1), dict(dwl=p["dwl"][:-1]), dict(wl_leafsoil=p["wl"][:-1])):
        with pytest.raises(canopy.CanopyInputError):
            canopy.derive({**p, **bad})
    # CanopyInputError is caught by handlers written for the reference's AssertionError as well as by ValueError handlers
    assert issubclass(canopy.CanopyInputError, AssertionError) and issubclass(canopy.CanopyInputError, ValueError)
    q = dict(p)
    del q["psi"]
    with pytest.raises(canopy.CanopyInputError, match="required key psi"):
        canopy.derive(q)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        canopy.derive({**p, "mu": 0.123})
        canopy.derive({**p, "wl_leafsoil": p["wl"] * 1.01})
        msgs = " ".join(str(x.message) for x in w)
    assert "not consistent with provided `psi`" in msgs and "appear to be incompatible" in msgs
