"""
``Model`` -- the reference's convenience driver (``crt1d/model.py:49-447``) on top of the MI355X solvers.

Same entry points and bookkeeping for the hot path: ``Model(scheme, nlayers, **p)``, ``update_p``,
``assign_scheme``, ``run(**solver_options)`` (``model.py:296-320``), ``calc_absorption()``
(``model.py:327-336,573-647``), ``out`` / ``out_extra`` / ``absorption``.  ``to_dataset()`` returns the reference's
output dataset as a plain container (``to_xr()`` converts it when xarray is installed); plots are out of scope.

Multi-column use (the reference's ``run_sensitivity`` is a ``NotImplementedError`` stub, ``model.py:650-664``):
:func:`run_columns` solves many independent canopies in one launch.
"""

import copy
import pprint
import traceback
import warnings

import numpy as np

from . import canopy
from .canopy import CANOPY_DESCRIPTION_KEYS, CanopyDescription
from .cases import load_default_case
from .solvers import AVAILABLE_SCHEMES, RET_KEYS_ALL_SCHEMES
from .variables import VMD

__all__ = ("Model", "Dataset", "run_columns")

_FALLBACK_SCHEME = "2s"
_SOLUTION_VARS = ("I_dr", "I_df_d", "I_df_u", "F")
_SCHEME_SUFFIX = "_scheme"


class Model:
    """One canopy, one scheme: holds the case, runs the scheme's solver on the GPU, keeps the results.

    State: ``_p`` (inputs + what :func:`crt1d_amd.canopy.derive` adds), ``scheme`` (registry entry), ``out`` (the four contract
    profiles), ``out_extra`` (everything else the scheme returned, keys suffixed ``_scheme``), ``absorption``."""

    required_input_keys = canopy.INPUT_KEYS
    _schemes = AVAILABLE_SCHEMES
    vmd = VMD

    def __init__(self, scheme="2s", nlayers=60, **p_kwargs):
        self.nlayers = nlayers
        self.p_default = load_default_case(nlayers=nlayers)
        self._p = copy.deepcopy(self.p_default)
        self._run_count = 0
        self.out, self.out_extra, self.absorption = {}, {}, None
        self.assign_scheme(scheme)
        self._check_inputs()  # the default case; a failing update below then has something valid to fall back to
        if p_kwargs:
            self.update_p(**p_kwargs)

    # ---- the case ---------------------------------------------------------------------------
    @property
    def p(self):
        """Not the parameters: a reminder (the reference does the same, ``model.py:108-116``), since edits to a returned dict
        would bypass validation."""
        print("Please update parameters using `.update_p()`! Changes to `.p` will not be stored!\n"
              "Extract (copy) the parameters using `.copy_p()` or summarize using `.print_p()`.")

    def copy_p(self):
        return copy.deepcopy(self._p)

    def print_p(self):
        with np.printoptions(precision=3, threshold=7):
            pprint.pprint(self._p, indent=1)

    @property
    def cd(self):
        return CanopyDescription(*(self._p[k] for k in CANOPY_DESCRIPTION_KEYS))

    def __repr__(self):
        return f"Model(scheme={self.scheme['name']!r}, psi={self._p['psi']:.4g})"

    def assign_scheme(self, scheme_name, *, verbose=False):
        """Select a registered scheme by ID.  An unknown ID is reported on stdout and replaced by '2s' (``model.py:157-168``)."""
        entry = AVAILABLE_SCHEMES.get(scheme_name)
        if entry is None:
            print(f"{scheme_name!r} is not a valid scheme name/ID!\nThe valid ones are: {', '.join(AVAILABLE_SCHEMES)}.\n"
                  "Defaulting to Dickinson-Sellers two-stream.\n")
            entry = AVAILABLE_SCHEMES[_FALLBACK_SCHEME]
        elif verbose:
            rule = "=" * 40
            print(f"\n\n{rule}\nscheme: {entry['name']}\n{'-' * 40}")
        self.scheme = entry
        return self

    def update_p(self, **kwargs):
        """Replace inputs.  All-or-nothing: the candidate case is validated as a whole and adopted only if that succeeds, otherwise
        a warning carries the traceback and the previous case stays (``model.py:173-203``).  Non-input keys are skipped with a
        warning."""
        accepted = {}
        for name, value in kwargs.items():
            if name in canopy.INPUT_KEYS:
                accepted[name] = value
            else:
                warnings.warn(f"{name!r} is not intended as an input and will be ignored")
        candidate = {**self._p, **accepted}
        try:
            candidate.update(canopy.derive(candidate))
        except Exception:
            warnings.warn(f"Updating parameters failed. Full traceback:\n\n{traceback.format_exc()}\nReverting.")
        else:
            self._adopt(candidate)
        return self

    def update_spectra(self, ds):
        """Inputs from a spectra container with ``I_dr, I_df, wl, dwl, tl, rl, rs`` (an ``xarray.Dataset`` in the reference, any
        mapping of arrays here)."""
        take = lambda k: np.asarray(getattr(ds[k], "values", ds[k]))  # noqa: E731
        renamed = {"I_dr0_all": "I_dr", "I_df0_all": "I_df", "wl": "wl", "dwl": "dwl", "leaf_t": "tl", "leaf_r": "rl", "soil_r": "rs",
                   "wl_leafsoil": "wl"}
        return self.update_p(**{ours: take(theirs) for ours, theirs in renamed.items()})

    def _adopt(self, case):
        self._p = case
        self.nlev, self.nwl = canopy.sizes(case)

    def _check_inputs(self):
        """(Re)derive ``lai_tot, lai_eff, dlai, zm, dz, mu, wle, K_b_fn, G, K_b`` from the inputs; raises on an unsolvable case
        (``model.py:222-294``)."""
        self._adopt({**self._p, **canopy.derive(self._p)})

    # ---- solve ------------------------------------------------------------------------------
    def run(self, **extra_solver_kwargs):
        """Call the scheme's solver with the arguments its signature names (``model.py:296-320``)."""
        self._check_inputs()
        solver, argnames = self.scheme["solver"], self.scheme["args"]
        returned = solver(**{name: self._p[name] for name in argnames}, **extra_solver_kwargs)
        for name, value in returned.items():
            if name in RET_KEYS_ALL_SCHEMES:
                self.out[name] = value
            else:
                self.out_extra[name + _SCHEME_SUFFIX] = value
        self._run_count += 1
        return self

    @property
    def out_all(self):
        return {**self.out, **self.out_extra}

    def _need_run(self, purpose):
        if not self._run_count:
            raise Exception(f"Must run the model {purpose}.")

    def calc_absorption(self):
        """Layerwise absorption (``model.py:573-647``), computed by the device epilogue kernel."""
        self._need_run("first")
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
        if not np.allclose(ab["aI_sl"] + ab["aI_sh"], ab["aI"]):  # the reference's sanity check, model.py:635
            raise AssertionError("sunlit + shaded absorption does not add up to the total")
        self.absorption = ab
        return self

    # ---- output container -------------------------------------------------------------------
    def _scheme_absorption_vars(self, nz):
        """The scheme's own absorption outputs (``aI*_scheme``): on levels or on layers, by their leading size."""
        from .variables import da_attrs

        for name, arr in self.out_extra.items():
            if not name.startswith("aI"):
                continue
            dims = {nz: ("z", "wl"), nz - 1: ("zm", "wl")}.get(arr.shape[0])
            if dims is None:
                raise ValueError("Scheme absorption output has too many or too few levels.")
            base = name[: -len(_SCHEME_SUFFIX)]
            try:
                yield name, (dims, arr, da_attrs(base))
            except KeyError:
                raise Exception(f"Scheme absorbance variable {base} not found in vmd.") from None

    def to_dataset(self, *, info=""):
        """The reference's output dataset (``model.py:338-447``) as a plain :class:`Dataset`: same coordinates (``z, wl, zm,
        wle``), data variables (solution, ``I_d``, grid, absorption, scheme extras named ``aI*_scheme``, geometry scalars),
        dims, attributes and global attributes.  No xarray needed; :meth:`to_xr` converts when it is installed."""
        from . import __version__
        from .variables import dv_tuple

        self._need_run("before creating the dataset")
        p, out = self._p, self.out
        values = {name: out[name] for name in _SOLUTION_VARS}
        values["I_d"] = out["I_dr"] + out["I_df_d"]
        values.update({name: np.asarray(p[name]) for name in ("dwl", "lai", "dlai")})
        values.update(self.absorption or {})  # standard absorption calculations (layer in-out)
        data_vars = {name: dv_tuple(name, v) for name, v in values.items()}
        data_vars.update(self._scheme_absorption_vars(np.asarray(p["z"]).size))
        scalars = {"psi": p["psi"], "sza": np.rad2deg(p["psi"]), "G": p["G"], "K_b": p["K_b"]}
        data_vars.update({name: dv_tuple(name, v) for name, v in scalars.items()})
        coords = {name: dv_tuple(name, np.asarray(p[name])) for name in ("z", "wl", "zm", "wle")}
        attrs = {"info": info, "crt1d_version": __version__}
        attrs.update({f"scheme_{k}": self.scheme[k] for k in ("name", "long_name", "short_name")})
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
