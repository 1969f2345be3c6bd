"""GPU: the drop-in layer (solve_* plugin functions, Model.run / calc_absorption), the epilogue kernels and the
kernel-selection paths, all through the C ABI."""
import numpy as np
import pytest

from conftest import load_golden, rel_profile_err

pytestmark = pytest.mark.gpu

TOL = {"2s": 1e-11, "g77": 1e-11, "bf": 1e-11, "bl": 1e-6, "n79": 1e-6, "zq": 1e-6, "zq_pa": 1e-6}


def _ref_style_kwargs(g):
    """Keyword arguments exactly as the reference's Model builds them: plain lambdas for G_fn / K_b_fn
    (cases.py:30, model.py:291) -> exercises the sampled-table path."""
    x = float(g["x"])
    G_fn = lambda psi_: np.sqrt(x**2 + np.tan(psi_) ** 2) / (x + 1.774 * (x + 1.182) ** -0.733) * np.cos(psi_)  # noqa: E731
    return dict(
        psi=float(g["psi"]), I_dr0_all=g["I_dr0_all"], I_df0_all=g["I_df0_all"], lai=g["lai"], leaf_t=g["leaf_t"], leaf_r=g["leaf_r"],
        soil_r=g["soil_r"], K_b_fn=lambda psi_: G_fn(psi_) / np.cos(psi_), G_fn=G_fn, mla=float(g["mla"]), clump=1.0,
    )


@pytest.mark.parametrize("scheme", ["2s", "4s", "n79", "zq", "bl", "g77", "bf", "zq_pa"])
def test_plugin_functions_with_plain_callables(scheme):
    from crt1d_amd import solvers

    g = load_golden("g1_default")
    sd = solvers.AVAILABLE_SCHEMES[scheme]
    kw = _ref_style_kwargs(g)
    before = {k: np.copy(v) for k, v in kw.items() if isinstance(v, np.ndarray)}
    sol = sd["solver"](**{k: kw[k] for k in sd["args"]})
    assert all(np.array_equal(before[k], kw[k]) for k in before)  # inputs are never mutated
    ref_keys = [k[len(scheme) + 2:] for k in g.files if k.startswith(scheme + "__")]
    assert sorted(sol) == sorted(ref_keys)
    for k, v in sol.items():
        if k == "rho_c":
            assert v == pytest.approx(float(g["bf__rho_c"]), rel=1e-14)
            continue
        assert isinstance(v, np.ndarray) and v.dtype == np.float64 and v.flags["C_CONTIGUOUS"] and v.flags["WRITEABLE"]
        assert v.shape == g[f"{scheme}__{k}"].shape
        if scheme == "4s":
            g5 = load_golden("g5_4s_tight")
            assert rel_profile_err(v, g5[f"4s_tol1e-11__{k}"]) <= 1e-8, k
        else:
            assert rel_profile_err(v, g[f"{scheme}__{k}"]) <= TOL[scheme], k


@pytest.mark.parametrize("scheme", ["2s", "4s", "n79", "zq", "bl", "g77", "bf", "zq_pa"])
def test_model_run_default_case(scheme):
    """Model(scheme).run() on the default canopy (BASELINE config 1: 1 profile x 107 bands x 60 levels)."""
    from crt1d_amd.model import Model

    g = load_golden("g1_default")
    m = Model(scheme, nlayers=60).run()
    assert sorted(m.out) == ["F", "I_df_d", "I_df_u", "I_dr"] and m._run_count == 1
    extra = [k[len(scheme) + 2:] for k in g.files if k.startswith(scheme + "__") and k.split("__")[1] not in m.out]
    assert sorted(m.out_extra) == sorted(f"{k}_scheme" for k in extra)
    for k, v in m.out.items():
        tol = 2e-4 if scheme == "4s" else TOL[scheme]
        assert rel_profile_err(v, g[f"{scheme}__{k}"]) <= tol, k


def test_model_options_and_errors():
    from crt1d_amd.model import Model

    g = load_golden("g7_options")
    from crt1d_amd.leaf_angle import GFunction

    m = Model("4s", nlayers=20)
    z = np.linspace(0.5, 20, 20)
    wl = np.linspace(0.4, 2.0, 6)
    m.update_p(lai=g["lai"], z=z, psi=float(g["psi"]), mla=float(g["mla"]), G_fn=GFunction(3, 2.0), leaf_r=g["leaf_r"], leaf_t=g["leaf_t"],
               soil_r=g["soil_r"], I_dr0_all=g["I_dr0_all"], I_df0_all=g["I_df0_all"], wl=wl, wl_leafsoil=wl, dwl=np.full(6, 0.3))
    m.run(mu_s=0.33998)
    for k, v in m.out.items():
        assert rel_profile_err(v, g[f"ellipsoidal_x2__4s_mus0.33998_tol1e-11__{k}"]) <= 1e-8, k
    m.assign_scheme("n79")
    with pytest.raises(ValueError):
        m.run(tau_d_method="simpson")
    with pytest.raises(TypeError):
        m.run(not_an_option=1)


def test_bonan_through_model_and_absorption(oracle):
    """The reference's own test (tests/test_n79.py): Model('n79', **p).run(tau_d_method='9sky').calc_absorption(); its
    Bonan table needs a download, so the golden is the reference's solver output for the same inputs, and the
    absorption is checked against the oracle's restatement of model.py:573-647 plus the test's own consistency
    (per-leaf-area sunlit/shaded absorption from calc_absorption == the scheme's aI_lsl / aI_lsh)."""
    from crt1d_amd import leaf_angle
    from crt1d_amd.model import Model

    g = load_golden("g2_bonan")
    wl = g["wl"]
    m = Model("n79", lai=g["lai"], z=g["z"], psi=float(g["psi"]), leaf_r=g["leaf_r"], leaf_t=g["leaf_t"], soil_r=g["soil_r"],
              I_dr0_all=g["I_dr0_all"], I_df0_all=g["I_df0_all"], wl=wl, wl_leafsoil=wl, dwl=g["dwl"], clump=1.0,
              G_fn=leaf_angle.G_spherical)
    m.run(tau_d_method="9sky").calc_absorption()
    for k in ("I_dr", "I_df_d", "I_df_u", "F"):
        assert rel_profile_err(m.out[k], g[f"n79_9sky__{k}"]) <= 1e-12, k
    cols = oracle.Columns([float(g["psi"])], g["lai"][None], g_kind=[1], g_param=[0.0])
    ab = oracle.calc_absorption(cols, {k: g[f"n79_9sky__{k}"][None] for k in ("I_dr", "I_df_d", "I_df_u")},
                                leaf_r=g["leaf_r"], leaf_t=g["leaf_t"])
    for k, v in m.absorption.items():
        ref = ab[k][0]
        assert np.max(np.abs(v - ref)) <= 1e-13 * max(1.0, np.abs(ref).max()), k
    f_sl, dlai = m.absorption["f_slm"], m.copy_p()["dlai"]
    y_sl = m.absorption["aI_sl"] / (f_sl * dlai)[:, None]
    y_sh = m.absorption["aI_sh"] / ((1 - f_sl) * dlai)[:, None]
    assert np.abs(y_sl - g["n79_9sky__aI_lsl"]).mean() < 1e-6  # the MAE bar of tests/test_n79.py:69-72
    assert np.abs(y_sh - g["n79_9sky__aI_lsh"]).mean() < 1e-6


@pytest.mark.parametrize("shape", [(37, 300, 60), (11, 38, 100), (9, 64, 33), (13, 20, 12), (5, 107, 61), (4, 512, 30), (3, 600, 20), (2, 1100, 9), (1, 36, 60),
                                   (7, 34, 2), (5, 62, 18), (6, 40, 17), (3, 52, 130), (4, 37, 40), (3, 63, 35)])
def test_epilogue_kernels_vs_oracle(oracle, shape):
    """crt_hip_absorb_f64 / crt_hip_absorb_bandsum_f64 on every kernel path: a column per half wave (nb <= 32), a wave per column with the
    lanes over layers (32 < nb <= 64, even; slabs of 16 layers: nz - 1 below, at and above multiples of 16), with the lanes over bands
    (nb <= 512), multi-wave workgroups with band slices (beyond); even nb -> tiled flat walk of the seven per-band outputs, odd nb ->
    lane per band."""
    import torch

    from crt1d_amd import batched, spectra, synth

    ncol, nb, nz = shape
    d = synth.make_columns(ncol, nb, nz, seed=8, uniform_dlai=False)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    sol = batched.solve("2s", cols, bands)
    w = spectra.band_weights(d["wle"])
    res = batched.absorb_bandsum(cols, bands, sol, torch.as_tensor(w).cuda())
    per = batched.absorb(cols, bands, sol)
    oc = oracle.Columns(d["psi"], d["lai"], mla=d["mla"], g_kind=d["g_kind"], g_param=d["g_param"])
    out = {k: sol[k].cpu().numpy() for k in ("I_dr", "I_df_d", "I_df_u")}
    ab = oracle.calc_absorption(oc, out, leaf_r=d["leaf_r"], leaf_t=d["leaf_t"])
    for k in batched.ABSORPTION_KEYS + ("laim", "f_slm"):
        ref = ab["f_slm" if k == "f_slm" else k]
        assert np.max(np.abs(per[k].cpu().numpy() - ref)) <= 1e-13 * np.abs(ref).max(), k
    for k in ("aI", "aI_sl", "aI_sh"):
        ref = ab[k] @ w.T
        assert np.max(np.abs(res[k].cpu().numpy() - ref)) <= 1e-12 * np.abs(ref).max(), k
    tot = res["totals"].cpu().numpy()
    np.testing.assert_allclose(tot[:, :, 0], (out["I_dr"][:, -1] + out["I_df_d"][:, -1]) @ w.T, rtol=1e-12)
    np.testing.assert_allclose(tot[:, :, 3], out["I_df_u"][:, 0] @ w.T, rtol=1e-12)
    # energy closure per column, solar group: in - reflected - (transmitted - soil reflected) == canopy absorption
    canopy = res["aI"].cpu().numpy()[:, :, 2].sum(axis=1)
    np.testing.assert_allclose(tot[:, 2, 0] - tot[:, 2, 1] - (tot[:, 2, 2] - tot[:, 2, 3]), canopy, rtol=1e-10)


@pytest.mark.parametrize("scheme", ["2s", "4s", "bl", "g77", "bf", "n79", "zq"])
@pytest.mark.parametrize("shape", [(33, 300, 60), (5, 107, 61), (9, 64, 13), (7, 100, 60), (3, 128, 7), (2, 512, 33), (2, 1024, 9), (130, 77, 5),
                                   (4, 300, 100), (3, 65, 3), (2, 300, 8), (2, 96, 12), (2, 200, 25)])
def test_tile_kernel_equals_direct_kernel(scheme, shape):
    """The LDS-tiled, line-aligned kernel and the direct-store kernel run the same per-lane arithmetic:
    results must be BITWISE equal for every shape class (odd nb, several columns per workgroup, ragged last tile...)."""
    import torch

    from crt1d_amd import _lib, batched, synth

    ncol, nb, nz = shape
    d = synth.make_columns(ncol, nb, nz, seed=3)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    a = batched.Plan(scheme, cols, bands)
    b = batched.Plan(scheme, cols, bands)
    for v in list(a.out.values()) + list(b.out.values()):
        v.fill_(float("nan"))
    a()
    b(flags=_lib.FLAG_DIRECT_STORES)
    torch.cuda.synchronize()
    for k in a.out:
        assert bool(torch.isfinite(a.out[k]).all()), k  # every element written
        assert torch.equal(a.out[k], b.out[k]), k


def test_plan_flags_and_strided_outputs():
    import torch

    from crt1d_amd import _lib, batched, synth

    d = synth.make_columns(64, 300, 60, seed=4)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    plan = batched.Plan("n79", cols, bands)
    full = {k: v.clone() for k, v in plan().items()}
    # new spectra, same geometry: skipping K0 must give the same answer as a full run
    bands2 = batched.Bands(bands.I_dr0 * 0.5, bands.I_df0 + 1.0, bands.leaf_r, bands.leaf_t, bands.soil_r)
    p2 = batched.Plan("n79", cols, bands2, workspace=plan.workspace)
    a = {k: v.clone() for k, v in p2(flags=_lib.FLAG_SKIP_PRECOMPUTE).items()}
    b = batched.solve("n79", cols, bands2)
    for k in a:
        assert torch.equal(a[k], b[k]), k
        assert not torch.equal(a[k], full[k])
    with pytest.raises(ValueError):
        batched.Plan("2s", cols, bands, workspace=torch.empty(16, dtype=torch.uint8, device="cuda"))


@pytest.mark.parametrize("scheme", ["2s", "4s", "bl", "g77", "bf", "n79", "zq", "zq_pa"])
@pytest.mark.parametrize("shape", [(21, 300, 60), (5, 107, 61), (9, 64, 13), (40, 6, 20), (3, 130, 100), (2, 40, 400), (1, 1500, 30)])
def test_f32_storage_variant(scheme, shape):
    """crt_hip_*_f32: float spectra in, float profiles out, fp64 arithmetic.  Feeding the SAME (float-representable)
    inputs to the f64 entry point and rounding its outputs to float must give the identical bits (config 5:
    'fp32 vs fp64 tolerance' -> the only difference is the final rounding, <= 2^-24 relative)."""
    import torch

    from crt1d_amd import batched, synth

    ncol, nb, nz = shape
    d = synth.make_columns(ncol, nb, nz, seed=17, uniform_dlai=(ncol % 2 == 1))
    cols = batched.Columns.from_host(d)
    b32 = batched.Bands.from_host({k: (d[k].astype(np.float32) if k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r") else d[k]) for k in d})
    assert b32.dtype == torch.float32
    if scheme == "zq_pa" and (nb < 16 or nb > 832):  # f32 zq_pa exists in the single-kernel form only (16 <= nb, LDS-limited above)
        with pytest.raises(RuntimeError, match="not supported"):
            batched.solve(scheme, cols, b32)
        return
    b64 = batched.Bands(*[None if t is None else t.double() for t in (b32.I_dr0, b32.I_df0, b32.leaf_r, b32.leaf_t, b32.soil_r)])
    s32 = batched.solve(scheme, cols, b32)
    s64 = batched.solve(scheme, cols, b64)
    for k in s32:
        assert s32[k].dtype == torch.float32 and s32[k].shape == s64[k].shape
        assert torch.equal(s32[k], s64[k].float()), k
    with pytest.raises(TypeError):
        batched.Bands(b32.I_dr0, b64.I_df0, b32.leaf_r, b32.leaf_t, b32.soil_r)


@pytest.mark.parametrize("scheme", ["2s", "n79"])
def test_dist_paths_on_the_hip_kernels(oracle, scheme):
    """crt1d_amd.dist with its default (HIP) compute functions, world size 1: both partitions give the same integrated
    results, equal to the oracle's.  (The multi-rank packing / all-reduce logic is covered by tests/test_dist_gloo.py.)"""
    import torch

    from crt1d_amd import batched, spectra, synth
    from crt1d_amd.dist import solve_sharded

    d = synth.make_columns(23, 300, 60, seed=12)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    w = torch.as_tensor(spectra.band_weights(d["wle"])).cuda()
    rc = solve_sharded(scheme, cols, bands, w, partition="column")
    rb = solve_sharded(scheme, cols, bands, w, partition="band")
    assert rc["columns"] == (0, 23) and rb["columns"] == (0, 23)
    for k in ("aI", "aI_sl", "aI_sh", "totals", "reflectance"):
        assert torch.equal(rc[k], rb[k]), k
    rt = solve_sharded(scheme, cols, bands, w, partition="band", column_tiles=4)  # column tiles (23 -> 6,6,6,5): same kernels
    assert len(rt["profiles"]) == 4 and rt["profiles"][3]["I_dr"].shape == (5, 60, 300)
    for k in ("aI", "aI_sl", "aI_sh", "totals", "reflectance"):
        assert torch.equal(rt[k], rb[k]), k
    rf = solve_sharded(scheme, cols, bands, w, partition="band", keep_profiles=False)  # fused kernel, no profiles
    assert rf["profiles"] is None
    for k in ("aI", "aI_sl", "aI_sh", "totals", "reflectance"):
        assert float((rf[k] - rc[k]).abs().max()) <= 1e-12 * float(rc["totals"].abs().max()), k
    oc = oracle.Columns(d["psi"], d["lai"], mla=d["mla"], g_kind=d["g_kind"], g_param=d["g_param"])
    kw = dict(I_dr0=d["I_dr0"], I_df0=d["I_df0"], leaf_r=d["leaf_r"], leaf_t=d["leaf_t"], soil_r=d["soil_r"])
    ref = oracle.SOLVERS[scheme](oc, **kw)
    ab = oracle.calc_absorption(oc, ref, leaf_r=d["leaf_r"], leaf_t=d["leaf_t"])
    wn = w.cpu().numpy()
    np.testing.assert_allclose(rc["aI"].cpu().numpy(), ab["aI"] @ wn.T, rtol=1e-9, atol=1e-12)
    refl = (ref["I_df_u"][:, -1] @ wn.T) / ((ref["I_dr"][:, -1] + ref["I_df_d"][:, -1]) @ wn.T)
    np.testing.assert_allclose(rc["reflectance"].cpu().numpy(), refl, rtol=1e-9)


@pytest.mark.parametrize("scheme", ["2s", "4s", "bl", "g77", "bf", "n79", "zq"])
@pytest.mark.parametrize("shape", [(19, 300, 60), (7, 107, 61), (5, 64, 13), (3, 30, 9), (2, 513, 100)])
def test_integrated_kernel_equals_solve_plus_epilogue(scheme, shape):
    """crt_hip_integrated_f64 (no profiles written) == crt_hip_<scheme>_f64 followed by crt_hip_absorb_bandsum_f64.
    The fused kernel differences Phi(k+1) - Phi(k) of band-summed net fluxes instead of summing per-band differences:
    same numbers up to rounding relative to the flux (1e-16 * |Phi| / |aI|)."""
    import torch

    from crt1d_amd import batched, spectra, synth

    ncol, nb, nz = shape
    d = synth.make_columns(ncol, nb, nz, seed=23, uniform_dlai=(nz % 2 == 0))
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    w = torch.as_tensor(spectra.band_weights(d["wle"])).cuda()
    fused = batched.solve_integrated(scheme, cols, bands, w)
    sol = batched.solve(scheme, cols, bands)
    ref = batched.absorb_bandsum(cols, bands, sol, w)
    flux = ref["totals"][:, :, 0].abs().amax()  # scale of the fluxes being differenced
    for k in ("aI", "aI_sl", "aI_sh", "totals"):
        err = float((fused[k] - ref[k]).abs().max() / flux)
        assert err < 1e-13, (k, err)


def test_columns_validate():
    import torch

    from crt1d_amd import batched, synth

    d = synth.make_columns(6, 8, 10)
    cols = batched.Columns.from_host(d).validate()
    bad = dict(d, lai=d["lai"][:, ::-1].copy())
    with pytest.raises(AssertionError):
        batched.Columns.from_host(bad).validate()
    bad = dict(d, g_kind=np.full(6, 6, dtype=np.int32))
    with pytest.raises(ValueError):
        batched.Columns.from_host(bad).validate()
    assert cols.ncol == 6 and cols.nz == 10


def test_model_to_dataset():
    """Output container of the reference (model.py:338-447): names, dims, attributes; to_xr needs xarray."""
    from crt1d_amd.model import Dataset, Model

    m = Model("n79", nlayers=20)
    with pytest.raises(Exception, match="Must run the model"):
        m.to_dataset()
    ds = m.run().calc_absorption().to_dataset(info="hello")
    assert isinstance(ds, Dataset)
    nz, nwl = 20, m.nwl
    assert ds.sizes == {"z": nz, "wl": nwl, "zm": nz - 1, "wle": nwl + 1}
    expected = {"I_dr", "I_df_d", "I_df_u", "F", "I_d", "dwl", "lai", "dlai", "aI", "aI_df", "aI_dr", "aI_sh", "aI_sl", "aI_df_sl",
                "aI_df_sh", "laim", "f_slm", "aI_lsl_scheme", "aI_lsh_scheme", "psi", "sza", "G", "K_b"}
    assert set(ds.data_vars) == expected and set(ds.coords) == {"z", "wl", "zm", "wle"}
    assert ds.data_vars["I_dr"][0] == ("z", "wl") and ds.data_vars["aI_lsl_scheme"][0] == ("zm", "wl")
    assert ds.data_vars["aI_lsl_scheme"][2] == {"long_name": "Absorbed irradiance by sunlit leaves", "units": "W m-2",
                                                  "units_long": "W (m2 leaf)-1"}
    assert ds.data_vars["F"][2] == {"long_name": "Actinic flux (binned)", "units": "W m-2"}
    np.testing.assert_array_equal(ds["I_d"], m.out["I_dr"] + m.out["I_df_d"])
    assert ds["sza"] == pytest.approx(20.0) and "wle" in ds and ds["wle"].shape == (nwl + 1,)
    assert ds.attrs["info"] == "hello" and ds.attrs["scheme_name"] == "n79" and ds.attrs["scheme_long_name"] == m.scheme["long_name"]
    try:
        import xarray  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            m.to_xr()
    else:
        assert set(m.to_xr().data_vars) == expected


# (tune key, value) settings that select each kernel family through crt_options.tune (per call; keys in csrc/crt_internal.hpp)
_CLOSED_PATHS = {"k_pipe (default)": {}, "k_tile": {2: 4}, "k_tile generic flush": {2: 4 | 2}, "k_pipe generic flush": {2: 2},
                 "k_pipe 1 store wave, T=2": {3: 1, 4: 2}, "k_pipe 4 store waves, T=8": {3: 4, 4: 8}}
_TRI_PATHS = {"default": {}, "k_tri_tile": {10: 1}, "k_tri_tile M8 T8": {10: 1, 8: 8, 9: 8}, "double-buffer pipeline": {10: 2},
              "register-staged pipeline": {10: 3}, "generic-flush pipeline": {10: 4}, "pipeline M16 T4, 2 store waves": {8: 16, 9: 4, 11: 2}}


@pytest.mark.parametrize("scheme", ["2s", "4s", "bl", "g77", "bf", "n79", "zq"])
@pytest.mark.parametrize("shape", [(23, 300, 60), (9, 107, 61), (6, 64, 13), (5, 128, 60), (3, 600, 33), (4, 255, 100), (7, 300, 7),
                                   (9, 38, 100), (7, 37, 60), (130, 36, 61), (5, 16, 30), (6, 21, 12),  # + the narrow band shards
                                   (40, 12, 60), (33, 6, 20), (21, 11, 33), (50, 4, 9), (17, 14, 100)])  # + very narrow spectra
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_every_kernel_family_gives_the_same_bits(scheme, shape, dtype):
    """The wave-specialised pipelines (double-buffered, register-staged, generic flush), the all-waves tile kernels and the
    per-wave / direct kernels share the per-lane arithmetic; whichever the heuristics pick, the outputs are BITWISE equal."""
    import torch

    from crt1d_amd import _lib, batched, synth

    ncol, nb, nz = shape
    d = synth.make_columns(ncol, nb, nz, seed=13, uniform_dlai=(ncol % 2 == 1))
    if dtype == "f32":
        d = {k: (v.astype(np.float32) if k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r") else v) for k, v in d.items()}
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    ref = batched.Plan(scheme, cols, bands)
    ref(flags=_lib.FLAG_DIRECT_STORES)
    paths = _TRI_PATHS if scheme in ("n79", "zq") else _CLOSED_PATHS
    names = set()
    for name, tune in paths.items():
        p = batched.Plan(scheme, cols, bands, tune=tune)  # the overrides travel with the plan's calls: no global state to restore
        for v in p.out.values():
            v.fill_(float("nan"))
        p()
        names.add(p.last_kernel())
        torch.cuda.synchronize()
        for k in p.out:
            assert bool(torch.isfinite(p.out[k]).all()), (name, k)
            assert torch.equal(p.out[k], ref.out[k]), (name, k)
    if nb >= 16:  # (below 10 bands the tridiagonal schemes have the per-wave kernel only, whatever the settings say)
        assert len(names) >= 2, names  # the settings really selected different kernels / configurations


def test_plan_placement_auto():
    """placement="auto" only changes WHERE the output arrays live (crt_hip_buffer_alloc_set: 512 MB physical chunks whose memory
    classes are interleaved across the arrays of the set): same results, a report of the classes, no effect on small problems or
    caller-provided buffers."""
    import torch

    from crt1d_amd import batched, synth

    d = synth.make_columns(2400, 300, 60, seed=6)  # 4 x 346 MB of outputs: above the 1 GB threshold of the set allocator
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    ref = batched.Plan("2s", cols, bands, placement="none")
    assert ref.placement_report is None
    ref()
    p = batched.Plan("2s", cols, bands, placement="auto")
    rep = p.placement_report
    assert rep is not None and set(rep["classes"]) == set(ref.out)
    assert all(isinstance(v, str) and len(v) == 1 and v in "XYZ?" for v in rep["classes"].values())  # one 512 MB chunk per array
    for v in p.out.values():
        assert v.data_ptr() % (2 << 20) == 0
        v.fill_(float("nan"))
    p()
    torch.cuda.synchronize()
    for k in ref.out:
        assert torch.equal(p.out[k], ref.out[k]), k
    st = batched.buffer_stats()
    assert st["chunks_created"] >= 4 and st["probes"] >= 1 and 1 <= st["classes_seen"] <= 3
    small = batched.Plan("2s", batched.Columns.from_host(synth.make_columns(8, 64, 10)), batched.Bands.from_host(synth.make_columns(8, 64, 10)),
                         placement="auto")
    assert small.placement_report is None
    assert batched.Plan("2s", cols, bands, out=ref.out, placement="auto").placement_report is None  # caller's buffers are kept
    with pytest.raises(ValueError):
        batched.Plan("2s", cols, bands, placement="best")


def test_device_buffer_set_lifecycle():
    """crt_hip_buffer_alloc_set / _free / _trim behind batched.device_buffers: normal torch tensors as far as the kernels and torch
    care; the chunks of freed buffers are reused; no virtual range is ever mapped twice (the ROCm 7.2 re-mapping hazard,
    csrc/buffers.hip), checked through the data: a live buffer keeps its contents while others are freed and re-allocated."""
    import gc

    import torch

    from crt1d_amd import _lib, batched, synth

    t = batched.device_buffer((3, 5, 7))
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.data_ptr() % (2 << 20) == 0
    t.fill_(2.5)
    assert float(t.sum()) == 2.5 * 105
    assert batched.buffer_classes(t) in ("X", "Y", "Z", "?") and batched.buffer_classes(torch.empty(3, device="cuda")) is None
    # a set of three arrays of 1.2 chunks each; the keeper holds a pattern across both of its chunks
    n = (600 << 20) // 8
    keep, b1, b2 = batched.device_buffers([(n,), (n,), (n,)])
    keep.copy_(torch.arange(n, dtype=torch.float64, device="cuda"))
    b1.fill_(1.0)
    b2.fill_(2.0)
    cls_keep = batched.buffer_classes(keep)
    assert len(cls_keep) == 2
    seen = {b1.data_ptr(), b2.data_ptr(), keep.data_ptr()}
    for rnd in range(3):  # free and re-allocate around the keeper: new virtual ranges every time, chunks reused from the pool
        del b1, b2
        gc.collect()
        b1, b2 = batched.device_buffers([(n,), (n,)])
        assert b1.data_ptr() not in seen and b2.data_ptr() not in seen
        seen |= {b1.data_ptr(), b2.data_ptr()}
        b1.fill_(10.0 + rnd)
        b2.fill_(20.0 + rnd)
        torch.cuda.synchronize()
        assert float(b1[0]) == 10.0 + rnd and float(b1[-1]) == 10.0 + rnd and float(b2[n // 2]) == 20.0 + rnd
        assert bool((keep[:: 4097] == torch.arange(0, n, 4097, dtype=torch.float64, device="cuda")).all())
    assert float(keep[n - 1]) == n - 1
    st0 = batched.buffer_stats()
    # the solve kernels run on such buffers like on any other memory
    d = synth.make_columns(2000, 128, 40, seed=2)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    shapes = [(2000, 40, 128)] * 4 + [(2000, 39, 128)] * 2
    out = dict(zip(batched.OUT_KEYS["n79"], batched.device_buffers(shapes)))
    got = batched.Plan("n79", cols, bands, out=out)()
    ref = batched.Plan("n79", cols, bands, placement="none")()
    torch.cuda.synchronize()
    for k in ref:
        assert torch.equal(got[k], ref[k]), k
    f32 = batched.device_buffer((4, 4), dtype=torch.float32)
    assert f32.dtype == torch.float32
    del t, out, got, f32, b1, b2, keep
    gc.collect()  # frees the buffers (crt_hip_buffer_free) without error
    st1 = batched.buffer_stats()
    assert st1["free_chunks"] >= 1
    _lib.check(_lib.load().crt_hip_buffer_trim(), "crt_hip_buffer_trim")
    st2 = batched.buffer_stats()
    assert st2["free_chunks"] == 0 and st2["chunks_released"] > st0["chunks_released"]
    with pytest.raises(ValueError):
        _lib.check(_lib.load().crt_hip_buffer_free(12345), "crt_hip_buffer_free")


def test_buffer_pool_retention_cap_and_availability():
    """The chunks of a freed set go back to the per-device pool only up to its retention cap (8 GB by default; ADVICE round 2: a dropped
    plan must not leave tens of GB where no other allocator can reach them); the rest returns to the driver inside crt_hip_buffer_free.
    A following request larger than what is FREE but smaller than free + pool must succeed (the budget counts the pool's memory), and a
    request that cannot be met hands everything back instead of sitting on it (the caller falls back to torch.empty, which needs it)."""
    import gc

    import torch

    from crt1d_amd import _lib, batched

    lib = _lib.load()
    lib.crt_hip_buffer_trim()
    gc.collect()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < (120 << 30):
        pytest.skip("needs 120 GB of free device memory")
    ballast = torch.empty(free - (64 << 30), dtype=torch.uint8, device="cuda")  # leave 64 GB
    n8 = lambda gb: (int(gb * (1 << 30)) // 8,)  # noqa: E731
    a = batched.device_buffers([n8(10)] * 3)       # 30 GB
    a[0][:16] = 1.0
    held, _ = torch.cuda.mem_get_info()
    del a
    gc.collect()                                   # -> at most 16 chunks (8 GB) stay in the pool, >= 22 GB are free again
    assert 1 <= batched.buffer_stats()["free_chunks"] <= 16
    back, _ = torch.cuda.mem_get_info()
    assert back - held >= (21 << 30), (held, back)
    big = torch.empty(20 << 30, dtype=torch.uint8, device="cuda")  # ... and torch's allocator can have them
    del big
    torch.cuda.empty_cache()
    batched.set_pool_retention(40 << 30)           # a larger cap for the second half of the test
    a = batched.device_buffers([n8(10)] * 3)
    del a
    gc.collect()                                   # -> 60 chunks in the pool, ~34 GB free
    assert batched.buffer_stats()["free_chunks"] >= 60
    b = batched.device_buffers([n8(11)] * 4)       # 44 GB: more than is free, less than free + pool
    b[3][-16:] = 2.0
    torch.cuda.synchronize()
    assert float(b[3][-1]) == 2.0
    with pytest.raises(RuntimeError):              # 4 x 20 GB cannot fit beside the 44 GB: fails ...
        batched.device_buffers([n8(20)] * 4)
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after >= (64 - 44 - 8) << 30       # ... without keeping what it gathered
    batched.set_pool_retention(8 << 30)            # back to the default: applied at once
    del b, ballast
    gc.collect()
    assert batched.buffer_stats()["free_chunks"] <= 16
    batched.trim_buffers()
    assert batched.buffer_stats()["free_chunks"] == 0
    torch.cuda.empty_cache()


def test_device_exp_and_sincos_accuracy():
    """The kernels' own exp / sincos (csrc/crt_internal.hpp: fexp -- 18 instructions, scalar-register coefficients; fast_sincos) against
    the host's libm in extended precision (numpy longdouble), over the arguments the schemes produce: exp <= 1 ulp (1.5 allowed) for
    -745 < x < 709 incl. the underflow range, exact at 0; sin / cos <= 3e-16 absolute for |x| <= 1e4."""
    import torch

    from crt1d_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-700, 700, 200000), rng.uniform(-40, 2, 400000), rng.uniform(-1e-3, 1e-3, 50000),
                        np.array([0.0, -0.0, 1.0, -1.0, 709.0, -708.0, -745.0, -800.0, 0.5 * np.log(2), -0.5 * np.log(2)])])
    xt = torch.as_tensor(x).cuda()
    e, sn, cs = torch.empty_like(xt), torch.empty_like(xt), torch.empty_like(xt)
    _lib.check(lib.crt_hip_probe_math_f64(xt.data_ptr(), xt.numel(), e.data_ptr(), sn.data_ptr(), cs.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "crt_hip_probe_math_f64")
    torch.cuda.synchronize()
    e = e.cpu().numpy()
    ref = np.exp(x.astype(np.longdouble))
    normal = (x > -708.0) & (x < 709.0)
    ulp = np.spacing(np.abs(ref[normal]).astype(np.float64))
    err = np.abs(e[normal].astype(np.longdouble) - ref[normal]) / ulp
    assert float(err.max()) <= 1.5, float(err.max())
    assert e[x == 0.0].tolist() == [1.0, 1.0]  # exp(+-0) = 1 exactly: I_dr at the canopy top is I_dr0 bit for bit
    assert e[-3] == 0.0 and 0.0 <= e[-4] < 1e-300  # gradual underflow instead of a NaN
    xs = np.concatenate([rng.uniform(-1e4, 1e4, 200000), rng.uniform(-7, 7, 200000)])
    xt = torch.as_tensor(xs).cuda()
    e, sn, cs = torch.empty_like(xt), torch.empty_like(xt), torch.empty_like(xt)
    _lib.check(lib.crt_hip_probe_math_f64(xt.data_ptr(), xt.numel(), e.data_ptr(), sn.data_ptr(), cs.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "crt_hip_probe_math_f64")
    torch.cuda.synchronize()
    xl = xs.astype(np.longdouble)
    assert float(np.abs(sn.cpu().numpy() - np.sin(xl)).max()) <= 3e-16
    assert float(np.abs(cs.cpu().numpy() - np.cos(xl)).max()) <= 3e-16
