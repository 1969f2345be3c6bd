"""
Solver plugins of the MI355X path, registered exactly like the reference's (``crt1d/solvers/__init__.py``):
every ``_solve_<id>.py`` module provides ``solve_<id>`` (keyword-only arguments), ``short_name``, ``long_name``;
required arguments must be canopy-radiation-state input keys, defaulted keyword-only arguments are the scheme's
options.  ``AVAILABLE_SCHEMES[id]`` has the same fields (``name, module_name, short_name, long_name, solver,
args, options``), so ``Model.run`` (``crt1d/model.py:305-310``) can dispatch to these unchanged.

Each ``solve_*`` can also be called on its own with the reference's keyword arguments.
"""

import inspect
import warnings
from importlib import import_module
from pathlib import Path

from ..variables import VMD as _vmd

__all__ = ["AVAILABLE_SCHEMES", "RET_KEYS_ALL_SCHEMES", "CANOPY_RAD_STATE_INPUT_KEYS"]

CANOPY_RAD_STATE_INPUT_KEYS = list(_vmd.intent("in"))
RET_KEYS_ALL_SCHEMES = ["I_dr", "I_df_d", "I_df_u", "F"]
assert all(k in _vmd.intent("out") for k in RET_KEYS_ALL_SCHEMES)


def _discover():
    here = Path(__file__).parent
    return {p.stem[len("_solve_"):]: p.stem for p in sorted(here.glob("_solve_*.py"))}


def _build_registry():
    schemes = {}
    for name, module_name in _discover().items():
        module = import_module(f".{module_name}", package=__name__)
        solver = getattr(module, f"solve_{name}")
        long_name = getattr(module, "long_name", "")
        if not long_name:
            warnings.warn(f"`long_name` not defined for solver module {module_name!r}")
        spec = inspect.getfullargspec(solver)
        defaults = spec.kwonlydefaults or {}
        args = [k for k in spec.kwonlyargs if k not in defaults]
        invalid = [k for k in args if k not in CANOPY_RAD_STATE_INPUT_KEYS]
        if invalid:
            warnings.warn(
                f"Some arguments for scheme {name!r} not compatible with the expected:\n"
                f"  {', '.join(CANOPY_RAD_STATE_INPUT_KEYS)}\n"
                f"As a result, {name!r} will not be loaded.\nInvalid keys:\n  {', '.join(invalid)}"
            )
            continue
        schemes[name] = dict(
            module_name=module_name, name=name, short_name=getattr(module, "short_name", name), long_name=long_name,
            solver=solver, args=args, options=list(defaults),
        )
    return schemes


AVAILABLE_SCHEMES = _build_registry()
"""scheme id -> dict(name, module_name, short_name, long_name, solver, args, options)"""

for _d in AVAILABLE_SCHEMES.values():
    globals()[_d["solver"].__name__] = _d["solver"]
    __all__.append(_d["solver"].__name__)
