"""Row a12 of SURVEY section 8 complete: everything ``crt1d.diagnostics.band`` returns (``crt1d/diagnostics.py:39-108``) -- the
band-integrated LEVEL profiles of I_dr, I_df_d, I_df_u, F, I_d and all seven absorption sums, in W m-2 and as photon flux density
(``calc_PFD``, ``:19-36, :92-104``) -- from the device: ``crt_hip_absorb_bandsum2_f64`` (profiles read once), ``crt_hip_integrated2_f64``
(no profile written), ``crt1d_amd.diagnostics.band`` (a single Model's dataset).  Pinned on ``tests/golden/g10_band_profiles.npz``: values
formed with the REFERENCE's own ``_x_frac_in_bounds`` / ``e_wl_umol`` on the reference's profiles (``oracle/gen_golden.py`` g10)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

LEVEL_KEYS = ("I_dr", "I_df_d", "I_df_u", "F", "I_d")
ABS_KEYS = ("aI", "aI_dr", "aI_df", "aI_sl", "aI_sh", "aI_df_sl", "aI_df_sh")


def _default_case(torch, g1):
    from crt1d_amd import batched

    dev = "cuda"
    t = lambda a: torch.as_tensor(np.atleast_1d(np.asarray(a, dtype=np.float64))).to(dev)  # noqa: E731
    cols = batched.Columns(psi=t(g1["psi"]), lai=t(g1["lai"])[None, :], g_kind=torch.tensor([4], dtype=torch.int32, device=dev),
                           g_param=t(g1["x"]), mla=t(g1["mla"]))
    bands = batched.Bands(t(g1["I_dr0_all"]), t(g1["I_df0_all"]), t(g1["leaf_r"]), t(g1["leaf_t"]), t(g1["soil_r"]))
    return cols, bands


def _rel(got, ref):
    return float(np.abs(got - ref).max() / np.abs(ref).max())


@pytest.mark.parametrize("scheme", ["2s", "n79", "zq"])
@pytest.mark.parametrize("pfd", [False, True])
def test_bandsum2_on_reference_profiles_vs_g10(scheme, pfd):
    """The epilogue kernel on the REFERENCE's profiles (g1) against the reference's band sums (g10): <= 1e-12 of each variable's maximum."""
    import torch

    from crt1d_amd import batched, spectra

    g1, g10 = load_golden("g1_default"), load_golden("g10_band_profiles")
    cols, bands = _default_case(torch, g1)
    sol = {k: torch.as_tensor(g1[f"{scheme}__{k}"])[None].cuda() for k in ("I_dr", "I_df_d", "I_df_u")}
    names = [str(n) for n in g10["band_names"]]
    w = spectra.band_weights(g10["wle"], names, wl=g10["wl"], pfd=pfd)
    if not pfd:
        assert np.array_equal(w, g10["w"])  # the weights themselves: bit for bit the reference's
    res = batched.absorb_bandsum(cols, bands, sol, torch.as_tensor(w).cuda(), profiles=True)
    res = {k: v[0].cpu().numpy() for k, v in res.items()}
    sfx = "__band_pfd" if pfd else "__band"
    for k in LEVEL_KEYS:
        assert _rel(res[k].T, g10[f"{scheme}__{k}{sfx}"]) <= 1e-12, k
    ab = batched.absorption_from_bandsums(res)
    scale = np.abs(g10[f"{scheme}__aI{sfx}"]).max()
    for k in ABS_KEYS:
        assert np.abs(ab[k].T - g10[f"{scheme}__{k}{sfx}"]).max() <= 1e-12 * scale, k


@pytest.mark.parametrize("scheme,tol", [("2s", 1e-10), ("n79", 1e-6), ("zq", 1e-6), ("4s", 2e-4)])
def test_integrated2_vs_g10(scheme, tol):
    """The fused path (solve + band integration, nothing written but the sums) against the reference's band sums: bounded by the solve's
    own parity with the reference (QUADPACK error for n79 / zq, the stock BVP tolerance for 4s; tests/test_gpu_parity.py)."""
    import torch

    from crt1d_amd import batched, spectra

    g1, g10 = load_golden("g1_default"), load_golden("g10_band_profiles")
    cols, bands = _default_case(torch, g1)
    names = [str(n) for n in g10["band_names"]]
    w = torch.as_tensor(spectra.band_weights(g10["wle"], names)).cuda()
    plan = batched.IntegratedPlan(scheme, cols, bands, w, profiles=True)
    res = {k: v[0].cpu().numpy() for k, v in plan().items()}
    assert "level profiles" in plan.lib.crt_hip_last_kernel().decode()
    for k in LEVEL_KEYS:
        assert _rel(res[k].T, g10[f"{scheme}__{k}__band"]) <= tol, k
    ab = batched.absorption_from_bandsums(res)
    scale = np.abs(g10[f"{scheme}__aI__band"]).max()
    for k in ABS_KEYS:
        assert np.abs(ab[k].T - g10[f"{scheme}__{k}__band"]).max() <= tol * scale, k


@pytest.mark.parametrize("shape", [(37, 300, 60), (11, 38, 100), (13, 20, 12), (5, 107, 61), (4, 512, 30), (3, 600, 20), (2, 1100, 9), (7, 34, 2), (3, 63, 35)])
@pytest.mark.parametrize("ngroup", [1, 3, 4])
def test_level_profiles_vs_oracle_all_kernel_paths(oracle, shape, ngroup):
    """profiles=True on every width class (a wave per column up to 512 bands, band slices beyond) and group count, PFD weights in the last
    group, against the oracle's band_profiles; the absorption sums are what the plain call returns (bitwise where both run the same kernel;
    narrow spectra take the lanes-over-layers / half-wave kernels without the profiles: another summation order); the fused kernel agrees."""
    import torch

    from crt1d_amd import batched, spectra, synth

    ncol, nb, nz = shape
    d = synth.make_columns(ncol, nb, nz, seed=8, uniform_dlai=False)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    sol = batched.solve("2s", cols, bands)
    names = ("PAR", "NIR", "solar", "PAR")[:ngroup]
    w = spectra.band_weights(d["wle"], names)
    if ngroup == 4:
        w[3] = spectra.band_weights(d["wle"], ("PAR",), pfd=True)[0]
    wt = torch.as_tensor(w).cuda()
    res = batched.absorb_bandsum(cols, bands, sol, wt, profiles=True)
    plain = batched.absorb_bandsum(cols, bands, sol, wt)
    flux0 = float(plain["totals"][:, :, 0].abs().amax())
    for k in ("aI", "aI_sl", "aI_sh", "totals"):
        if 48 < nb:
            assert torch.equal(res[k], plain[k]), k
        else:
            assert float((res[k] - plain[k]).abs().max()) <= 1e-13 * flux0, k
    oc = oracle.Columns(d["psi"], d["lai"], mla=d["mla"], g_kind=d["g_kind"], g_param=d["g_param"])
    out = {k: sol[k].cpu().numpy() for k in ("I_dr", "I_df_d", "I_df_u", "F")}
    ab = oracle.calc_absorption(oc, out, leaf_r=d["leaf_r"], leaf_t=d["leaf_t"])
    prof = dict(out, I_d=out["I_dr"] + out["I_df_d"])
    for k in LEVEL_KEYS:
        ref = prof[k] @ w.T
        assert _rel(res[k].cpu().numpy(), ref) <= 1e-12, k
    ref = ab["aI_dr"] @ w.T
    assert np.abs(res["aI_dr"].cpu().numpy() - ref).max() <= 1e-12 * np.abs(ab["aI"] @ w.T).max()
    if nb <= 1024:
        fused = batched.IntegratedPlan("2s", cols, bands, wt, profiles=True)()
        flux = float(res["totals"][:, :, 0].abs().amax())
        for k in batched.BANDSUM_KEYS + batched.PROFILE_KEYS:
            assert float((fused[k] - res[k]).abs().max()) <= 1e-13 * flux, k


@pytest.mark.parametrize("scheme", ["n79", "zq", "g77"])
def test_integrated_profiles_all_schemes(scheme):
    """crt_hip_integrated2_f64 with level profiles == solve + crt_hip_absorb_bandsum2_f64, tridiagonal and closed-form kernels."""
    import torch

    from crt1d_amd import batched, spectra, synth

    d = synth.make_columns(9, 107, 61, seed=4, uniform_dlai=False)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    wt = torch.as_tensor(spectra.band_weights(d["wle"])).cuda()
    fused = batched.IntegratedPlan(scheme, cols, bands, wt, profiles=True)()
    ref = batched.absorb_bandsum(cols, bands, batched.solve(scheme, cols, bands), wt, profiles=True)
    flux = float(ref["totals"][:, :, 0].abs().amax())
    for k in batched.BANDSUM_KEYS + batched.PROFILE_KEYS:
        assert float((fused[k] - ref[k]).abs().max()) <= 1e-13 * flux, k


def test_diagnostics_band_on_a_model_dataset():
    """crt1d_amd.diagnostics.band(Model.to_dataset()) == the reference's band() on its own run (g10), names, dims and the photon-flux
    variants included -- and the reference's naming quirk: "F".replace("I", "PFD") is "F", so with calc_PFD its F holds the PFD version."""
    from crt1d_amd import diagnostics
    from crt1d_amd.model import Model

    g10 = load_golden("g10_band_profiles")
    m = Model("2s", nlayers=60).run().calc_absorption()
    ds = m.to_dataset()
    for gi, name in enumerate(str(n) for n in g10["band_names"]):
        b = diagnostics.band(ds, band_name=name)
        assert "wl" not in b.coords and b.attrs["band_name"] == name
        for k in LEVEL_KEYS + ABS_KEYS:
            ref = g10[f"2s__{k}__band"][gi]
            assert b[k].shape == ref.shape and _rel(b[k], ref) <= 1e-10, (name, k)
        assert name in b.data_vars["I_dr"][2]["long_name"]
    b = diagnostics.band(ds, band_name="PAR", calc_PFD=True)
    for k in LEVEL_KEYS + ABS_KEYS:
        ref = g10[f"2s__{k}__band_pfd"][0]
        assert _rel(b[k.replace("I", "PFD")], ref) <= 1e-10, k
    assert b.data_vars["PFD_dr"][2]["units"] == "μmol photons m-2 s-1"
    assert _rel(b["I_dr"], g10["2s__I_dr__band"][0]) <= 1e-10  # the W m-2 version is still there under its own name


def test_smear_tuv_vs_reference_fixture():
    """crt_hip_smear_tuv_f64 against the reference's smear_tuv on seeded random spectra (g10; bins beyond the data on both sides):
    identical arithmetic in identical order -> identical bits."""
    import torch

    from crt1d_amd import spectra

    g10 = load_golden("g10_band_profiles")
    got = spectra.smear_tuv_batched(g10["smear_x"], torch.as_tensor(g10["smear_y"]).cuda(), g10["smear_bins"]).cpu().numpy()
    assert np.array_equal(got, g10["smear_out"])
    assert np.array_equal(spectra.smear_tuv(g10["smear_x"], g10["smear_y"][0], g10["smear_bins"]), g10["smear_out"][0])
