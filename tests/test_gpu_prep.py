"""GPU: input-side kernels through the C ABI -- spectral re-binning and leaf-area profiles."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize(
    "x,bins,expected",
    [
        (np.r_[0, 1, 2, 3, 4], (0, 2, 4), [1, 3]),
        (np.r_[0, 1, 2, 3, 4], (0, 1, 4), [0.5, 2.5]),
        (np.r_[0, 1, 2, 3, 4], (0, 1, 6), [0.5, 7.5 / (6 - 1)]),
        (np.r_[0, 1], (2, 3), [0]),
    ],
)
def test_smear_tuv_known_answers(x, bins, expected):
    """The reference's own test vectors (tests/test_spectra.py:38-55), y = x."""
    from crt1d_amd import spectra

    np.testing.assert_allclose(spectra.smear_tuv(x, x, bins), expected)


@pytest.mark.parametrize("shared_grid", [True, False])
def test_smear_tuv_batched_bit_exact_vs_oracle(oracle, shared_grid):
    """Same operations in the same order as `_smear_tuv_1` -> identical bits, including bins that stick out of the x
    range on both sides, bins narrower than one x interval and bin edges that coincide with x values."""
    import torch

    from crt1d_amd import spectra

    rng = np.random.default_rng(21)
    nspec, nx = 7, 2151
    if shared_grid:
        x = 0.35 + 1e-3 * np.arange(nx)  # 1 nm leaf spectrum, 0.35-2.5 um
        xs = np.broadcast_to(x, (nspec, nx))
    else:
        xs = 0.3 + np.cumsum(rng.uniform(2e-4, 3e-3, (nspec, nx)), axis=1)
        x = xs
    y = rng.uniform(0, 1, (nspec, nx))
    bins = np.r_[0.1, 0.2, 0.3501, 0.3502, 0.35025, np.linspace(0.36, 2.6, 300), 2.7, 4.0]
    bins[10] = xs[0, 17]  # an edge exactly on a grid value
    got = spectra.smear_tuv_batched(x, y, bins).cpu().numpy()
    assert got.shape == (nspec, bins.size - 1)
    for s in range(nspec):
        ref = oracle.smear_tuv(xs[s], y[s], bins)
        assert np.array_equal(got[s], ref), (s, np.abs(got[s] - ref).max())
    # the docstring property (spectra.py:263-266): the re-binned spectrum keeps the trapezoidal integral
    inner = np.linspace(xs[0, 0], xs[0, -1], 41)
    g2 = spectra.smear_tuv_batched(x, y, inner).cpu().numpy()
    np.testing.assert_allclose((g2[0] * np.diff(inner)).sum(), np.trapezoid(y[0], xs[0]), rtol=1e-12)
    with pytest.raises(ValueError):
        spectra.smear_tuv_batched(np.zeros(5), torch.zeros(2, 6), bins)


def test_smear_tuv_empty_and_single_point():
    from crt1d_amd import spectra

    assert spectra.smear_tuv_batched(np.arange(4.0), np.zeros((0, 4)), np.r_[0.0, 1.0]).shape == (0, 1)
    assert spectra.smear_tuv_batched(np.arange(4.0), np.ones((2, 4)), np.r_[1.0]).shape == (2, 0)
    np.testing.assert_array_equal(spectra.smear_tuv(np.r_[1.0], np.r_[5.0], np.r_[0.0, 2.0]), [0.0])  # no trapezoid at all


def test_distribute_lai_beta_vs_reference():
    """tests/golden/g8_leaf_area.npz: outputs of the reference's leaf_area.distribute_lai_beta.  lai is bit-identical
    (numpy.linspace arithmetic); z / lad go through a different Beta-ppf algorithm than scipy's: <= 1e-12 relative."""
    from crt1d_amd import leaf_area

    g = load_golden("g8_leaf_area")
    for i in range(len(g["h_c"])):
        r = leaf_area.distribute_lai_beta(float(g["h_c"][i]), float(g["LAI"][i]), int(g["n"][i]), h_min=float(g["h_min"][i]))
        assert np.array_equal(r.lai, g[f"c{i}__lai"]), i
        np.testing.assert_allclose(r.z, g[f"c{i}__z"], rtol=1e-12, err_msg=str(i))
        np.testing.assert_allclose(r.lad, g[f"c{i}__lad"], rtol=1e-10, atol=1e-14, err_msg=str(i))
        assert r.z[0] == g["h_min"][i] and r.z[-1] == g["h_c"][i] and np.all(np.diff(r.z) > 0)


def test_distribute_lai_beta_batched_feeds_the_solver(oracle):
    """Profiles built on the device are valid solver input (strictly decreasing lai ending at 0) and agree with the oracle."""
    import torch

    from crt1d_amd import batched, leaf_area, synth

    rng = np.random.default_rng(2)
    ncol, nz = 300, 60
    h_c, LAI, h_min = rng.uniform(2, 40, ncol), rng.uniform(0.5, 8, ncol), rng.uniform(0, 1.5, ncol)
    prof = leaf_area.distribute_lai_beta_batched(h_c, LAI, nz, h_min=h_min)
    assert prof.lai.is_cuda and prof.lai.shape == (ncol, nz)
    for c in (0, 17, 299):
        lai, lad, z = oracle.distribute_lai_beta(h_c[c], LAI[c], nz, h_min[c])
        assert np.array_equal(prof.lai[c].cpu().numpy(), lai)
        np.testing.assert_allclose(prof.z[c].cpu().numpy(), z, rtol=1e-12)
        np.testing.assert_allclose(prof.lad[c].cpu().numpy(), lad, rtol=1e-10, atol=1e-14)
    d = synth.make_columns(ncol, 16, nz, seed=4)
    d["lai"] = prof.lai.cpu().numpy()
    cols = batched.Columns.from_host(d)
    cols.validate()
    sol = batched.solve("2s", cols, batched.Bands.from_host(d))
    assert bool(torch.isfinite(sol["F"]).all())
    assert leaf_area.distribute_lai_beta_batched(h_c, LAI, nz, want_lad=False).lad is None
    with pytest.raises(ValueError):
        leaf_area.distribute_lai_beta_batched(h_c, LAI[:5], nz)
