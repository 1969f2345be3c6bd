"""
``Model`` -- the reference's convenience driver (``crt1d/model.py:49-447``) on top of the MI355X solvers.

Same entry points and bookkeeping for the hot path: ``Model(scheme, nlayers, **p)``, ``update_p``,
``assign_scheme``, ``run(**solver_options)`` (``model.py:296-320``), ``calc_absorption()``
(``model.py:327-336,573-647``), ``out`` / ``out_extra`` / ``absorption``.  ``to_dataset()`` returns the reference's
output dataset as a plain container (``to_xr()`` converts it when xarray is installed); plots are out of scope.

Multi-column use (the reference's ``run_sensitivity`` is a ``NotImplementedError`` stub, ``model.py:650-664``):
:func:`run_columns` solves many independent canopies in one launch.
"""

import warnings
from collections import namedtuple
from copy import deepcopy

import numpy as np

from .cases import load_default_case
from .solvers import AVAILABLE_SCHEMES, RET_KEYS_ALL_SCHEMES
from .solvers.common import KbFunction
from .variables import VMD

__all__ = ("Model", "Dataset", "run_columns")

CANOPY_DESCRIPTION_KEYS = [
    "lai", "z", "dlai", "lai_tot", "lai_eff", "mla", "clump", "leaf_t", "leaf_r", "soil_r", "wl_leafsoil", "orient", "G_fn",
]
CanopyDescription = namedtuple("CanopyDescription", " ".join(CANOPY_DESCRIPTION_KEYS))


class Model:
    """A general class for testing 1-D canopy radiative transfer schemes (GPU-backed)."""

    required_input_keys = tuple(
        [k for k in CANOPY_DESCRIPTION_KEYS if k not in ("dlai", "lai_tot", "lai_eff")]
        + ["I_dr0_all", "I_df0_all", "wl", "dwl", "psi"]
    )
    _schemes = AVAILABLE_SCHEMES
    vmd = VMD

    def __init__(self, scheme="2s", nlayers=60, **p_kwargs):
        self.nlayers = nlayers
        self.p_default = load_default_case(nlayers=self.nlayers)
        self._p = deepcopy(self.p_default)
        self.assign_scheme(scheme)
        if p_kwargs:
            self.update_p(**p_kwargs)
        else:
            self._check_inputs()
        self._run_count = 0
        self.absorption = None
        self.out = {}
        self.out_extra = {}

    # ---- parameters -------------------------------------------------------------------------
    @property
    def p(self):
        print(
            "Please update parameters using `.update_p()`! Changes to `.p` will not be stored!\n"
            "Extract (copy) the parameters using `.copy_p()` or summarize using `.print_p()`."
        )

    def print_p(self):
        import pprint

        with np.printoptions(precision=3, threshold=7):
            pprint.PrettyPrinter(indent=1).pprint(self._p)

    def copy_p(self):
        return deepcopy(self._p)

    @property
    def cd(self):
        return CanopyDescription(**{k: v for k, v in self._p.items() if k in CANOPY_DESCRIPTION_KEYS})

    def __repr__(self):
        return f"Model(scheme={self.scheme['name']!r}, psi={self._p['psi']:.4g})"

    def assign_scheme(self, scheme_name, *, verbose=False):
        """Unknown names print a message and fall back to '2s', as the reference (``model.py:157-168``)."""
        try:
            self.scheme = AVAILABLE_SCHEMES[scheme_name]
            if verbose:
                print("\n\n" + "=" * 40 + f"\nscheme: {self.scheme['name']}\n" + "-" * 40)
        except KeyError:
            print(f"{scheme_name!r} is not a valid scheme name/ID!")
            print(f"The valid ones are: {', '.join(AVAILABLE_SCHEMES)}.")
            print("Defaulting to Dickinson-Sellers two-stream.\n")
            self.scheme = AVAILABLE_SCHEMES["2s"]
        return self

    def update_p(self, **kwargs):
        """Update parameters if validation passes; on any failure warn and revert (``model.py:173-203``)."""
        import traceback

        p0 = deepcopy(self._p)
        try:
            for k, v in kwargs.items():
                if k not in Model.required_input_keys:
                    warnings.warn(f"{k!r} is not intended as an input and will be ignored")
                    continue
                self._p[k] = v
            self._check_inputs()
        except Exception:
            warnings.warn(f"Updating parameters failed. Full traceback:\n\n{traceback.format_exc()}\nReverting.")
            self._p = p0
        return self

    def update_spectra(self, ds):
        """From any mapping with ``I_dr, I_df, wl, dwl, tl, rl, rs`` (an ``xarray.Dataset`` in the reference)."""
        g = lambda k: np.asarray(getattr(ds[k], "values", ds[k]))  # noqa: E731
        self.update_p(I_dr0_all=g("I_dr"), I_df0_all=g("I_df"), wl=g("wl"), dwl=g("dwl"), leaf_t=g("tl"), leaf_r=g("rl"),
                      soil_r=g("rs"), wl_leafsoil=g("wl"))
        return self

    def _check_inputs(self):
        """Derive ``lai_tot, lai_eff, dlai, zm, dz, mu, wle, K_b_fn, G, K_b`` and validate (``model.py:222-294``)."""
        p = self._p
        for key in Model.required_input_keys:
            if key not in p:
                raise Exception(f"required key {key} is not present. Set it using `update_p`.")
        lai = np.asarray(p["lai"], dtype=float)
        z = np.asarray(p["z"], dtype=float)
        dz = np.diff(z)
        assert z.size == lai.size
        self.nlev = lai.size
        assert z[-1] > z[0]  # z increasing
        assert lai[0] > lai[-1]  # LAI decreasing
        assert lai[-1] == 0
        p["lai_tot"] = lai[0]
        p["lai_eff"] = lai * p["clump"]
        dlai = lai[:-1] - lai[1:]
        p["dlai"] = dlai
        p["dlai_eff"] = dlai * p["clump"]
        p["zm"] = z[:-1] + 0.5 * dz
        p["dz"] = dz
        psi = p["psi"]
        if "mu" in p:
            if p["mu"] != np.cos(psi):
                warnings.warn("Provided `mu` not consistent with provided `psi`. `mu` will be updated based on the value of `psi`.")
        p["mu"] = np.cos(psi)
        wl_toc, wl_op = np.asarray(p["wl"]), np.asarray(p["wl_leafsoil"])
        assert wl_toc.size == wl_op.size
        if not np.allclose(wl_toc, wl_op):
            warnings.warn(
                "Provided wavelengths for optical props (`wl_leafsoil`) and toc BC (`wl`) appear to be incompatible:\n"
                f"`wl - wl_leafsoil`:\n{wl_toc - wl_op}"
            )
        self.nwl = wl_toc.size
        assert np.asarray(p["wl"]).size == np.asarray(p["dwl"]).size
        p["wle"] = np.r_[p["wl"][0] - 0.5 * p["dwl"][0], np.asarray(p["wl"]) + 0.5 * np.asarray(p["dwl"])]
        p["K_b_fn"] = KbFunction(p["G_fn"])  # lambda psi_: G_fn(psi_) / cos(psi_), model.py:291
        p["G"] = p["G_fn"](psi)
        p["K_b"] = p["K_b_fn"](psi)

    # ---- run --------------------------------------------------------------------------------
    def run(self, **extra_solver_kwargs):
        self._check_inputs()
        scheme = self.scheme
        p = self._p
        args = {k: p[k] for k in scheme["args"]}
        sol = scheme["solver"](**args, **extra_solver_kwargs)
        self.out.update({k: v for k, v in sol.items() if k in RET_KEYS_ALL_SCHEMES})
        self.out_extra.update({f"{k}_scheme": v for k, v in sol.items() if k not in RET_KEYS_ALL_SCHEMES})
        self._run_count += 1
        return self

    @property
    def out_all(self):
        return {**self.out, **self.out_extra}

    def calc_absorption(self):
        """Layerwise absorption (``model.py:573-647``), computed by the device epilogue kernel."""
        if self._run_count == 0:
            raise Exception("Must run the model first.")
        import torch

        from . import batched
        from .leaf_angle import G_TABLE, describe_G

        p = self._p
        dev = torch.device("cuda", torch.cuda.current_device())
        t = lambda a: torch.as_tensor(np.atleast_1d(np.asarray(a, dtype=np.float64))).to(dev)  # noqa: E731
        d = describe_G(p["G_fn"])
        kind, param = d if d is not None else (G_TABLE, 0.0)
        cols = batched.Columns(
            psi=t(p["psi"]), lai=t(p["lai"])[None, :], g_kind=torch.tensor([kind], dtype=torch.int32, device=dev), g_param=t(param),
            g_at_psi=t(p["G"]),
        )
        bands = batched.Bands(t(p["I_dr0_all"]), t(p["I_df0_all"]), t(p["leaf_r"]), t(p["leaf_t"]), t(p["soil_r"]))
        sol = {k: t(self.out[k])[None] for k in ("I_dr", "I_df_d", "I_df_u")}
        res = batched.absorb(cols, bands, sol)
        ab = {k: v[0].cpu().numpy() for k, v in res.items()}
        ab["f_slm"] = ab.pop("f_slm")
        assert np.allclose(ab["aI_sl"] + ab["aI_sh"], ab["aI"])  # sanity check, model.py:635
        self.absorption = ab
        return self

    def to_dataset(self, *, info=""):
        """The reference's output dataset (``model.py:338-447``) as a plain :class:`Dataset`: same coordinates (``z, wl, zm,
        wle``), data variables (solution, ``I_d``, grid, absorption, scheme extras named ``aI*_scheme``, geometry scalars),
        dims, attributes and global attributes.  No xarray needed; :meth:`to_xr` converts when it is installed."""
        from . import __version__
        from .variables import da_attrs, dv_tuple

        if self._run_count == 0:
            raise Exception("Must run the model before creating the dataset.")
        p = self._p
        out = self.out_all
        z, zm = np.asarray(p["z"]), np.asarray(p["zm"])
        data_vars = {
            "I_dr": dv_tuple("I_dr", out["I_dr"]),
            "I_df_d": dv_tuple("I_df_d", out["I_df_d"]),
            "I_df_u": dv_tuple("I_df_u", out["I_df_u"]),
            "F": dv_tuple("F", out["F"]),
            "I_d": dv_tuple("I_d", out["I_dr"] + out["I_df_d"]),
            "dwl": dv_tuple("dwl", np.asarray(p["dwl"])),
            "lai": dv_tuple("lai", np.asarray(p["lai"])),
            "dlai": dv_tuple("dlai", np.asarray(p["dlai"])),
        }
        for k, v in (self.absorption or {}).items():  # standard absorption calculations (layer in-out)
            data_vars[k] = dv_tuple(k, v)
        for name, arr in self.out_extra.items():  # the scheme's own absorption outputs
            if name[:2] != "aI":
                continue
            if arr.shape[0] == z.size:
                dims = ("z", "wl")
            elif arr.shape[0] == zm.size:
                dims = ("zm", "wl")
            else:
                raise ValueError("Scheme absorption output has too many or too few levels.")
            try:
                attrs = da_attrs(name[:-7])  # without the `_scheme` suffix
            except KeyError:
                raise Exception(f"Scheme absorbance variable {name[:-7]} not found in vmd.")
            data_vars[name] = (dims, arr, attrs)
        data_vars.update(psi=dv_tuple("psi", p["psi"]), sza=dv_tuple("sza", np.rad2deg(p["psi"])), G=dv_tuple("G", p["G"]),
                         K_b=dv_tuple("K_b", p["K_b"]))
        coords = {"z": dv_tuple("z", z), "wl": dv_tuple("wl", np.asarray(p["wl"])), "zm": dv_tuple("zm", zm),
                  "wle": dv_tuple("wle", np.asarray(p["wle"]))}
        attrs = {"info": info, "scheme_name": self.scheme["name"], "scheme_long_name": self.scheme["long_name"],
                 "scheme_short_name": self.scheme["short_name"], "crt1d_version": __version__}
        return Dataset(coords, data_vars, attrs)

    def to_xr(self, *, info=""):
        """``xarray.Dataset`` of the run (``model.py:338-447``); needs xarray, which the hot path itself does not."""
        return self.to_dataset(info=info).to_xarray()


class Dataset:
    """Minimal labelled container: ``coords`` / ``data_vars`` map a name to ``(dims, array, attrs)``; ``attrs`` are global."""

    def __init__(self, coords, data_vars, attrs):
        self.coords, self.data_vars, self.attrs = dict(coords), dict(data_vars), dict(attrs)
        sizes = {k: np.shape(v[1])[0] for k, v in self.coords.items()}
        for name, (dims, arr, _) in self.data_vars.items():
            if tuple(sizes[d] for d in dims) != np.shape(arr):
                raise ValueError(f"{name}: shape {np.shape(arr)} does not match dims {dims} {sizes}")
        self.sizes = sizes

    def __getitem__(self, name):
        return (self.data_vars[name] if name in self.data_vars else self.coords[name])[1]

    def __contains__(self, name):
        return name in self.data_vars or name in self.coords

    def keys(self):
        return list(self.coords) + list(self.data_vars)

    def to_xarray(self):
        try:
            import xarray as xr
        except ImportError as e:
            raise ImportError("xarray is not installed; use the Dataset returned by Model.to_dataset() directly") from e
        return xr.Dataset(coords=self.coords, data_vars=self.data_vars, attrs=self.attrs)


def run_columns(scheme, columns, **solver_options):
    """Solve many canopies at once.

    ``columns``: dict of host arrays as produced by :func:`crt1d_amd.synth.make_columns`
    (``psi (ncol,)``, ``lai (ncol,nz)``, ``mla``, ``g_kind``, ``g_param``, per-(column, band) spectra).
    Returns a dict of ``(ncol, nz, nb)`` CUDA tensors (no host copy)."""
    from . import batched

    cols = batched.Columns.from_host(columns)
    bands = batched.Bands.from_host(columns)
    return batched.solve(scheme, cols, bands, **solver_options)
