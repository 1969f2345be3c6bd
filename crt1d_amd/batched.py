"""
Batched (column x band) canopy-RT solves on PyTorch-ROCm tensors.

PyTorch is plumbing here: device memory, streams and (in :mod:`crt1d_amd.dist`)
``torch.distributed``.  All arithmetic happens in the hand-written gfx950 kernels of
``libcrt1d_hip.so``, reached through the C ABI with ``tensor.data_ptr()``.

Shapes: ``ncol`` columns, ``nb`` bands, ``nz`` interface levels.  Outputs are
``(ncol, nz, nb)`` (``(ncol, nz-1, nb)`` for n79's per-leaf-area absorption), bands contiguous --
the reference's ``(nz, nb)`` solver outputs (e.g. ``_solve_2s.py:45-49``) stacked over columns.
"""

import ctypes
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib

SCHEMES = tuple(_lib.SCHEME_IDS)

# output keys per scheme, in crt_outputs slot order (I_dr, I_df_d, I_df_u, F, x0, x1, x2)
OUT_KEYS = {
    "2s": ("I_dr", "I_df_d", "I_df_u", "F"),  # _solve_2s.py:158-163
    "4s": ("I_dr", "I_df_d", "I_df_u", "F"),  # _solve_4s.py:293
    "bl": ("I_dr", "I_df_d", "I_df_u", "F"),  # _solve_bl.py:93
    "n79": ("I_dr", "I_df_d", "I_df_u", "F", "aI_lsl", "aI_lsh"),  # _solve_n79.py:157-164
    "zq": ("I_dr", "I_df_d", "I_df_u", "F", "I_df_d_ss", "I_df_u_ss", "F_ss"),  # _solve_zq.py:221-229
    "g77": ("I_dr", "I_df_d", "I_df_u", "F", "aI_lsl", "aI_lsh", "aI_l"),  # _solve_g77.py:127-135
    "bf": ("I_dr", "I_df_d", "I_df_u", "F", "aI_lsl", "aI_lsh", "aI_l"),  # _solve_bf.py:144-153 (+ rho_c host-side)
    "zq_pa": ("I_dr", "I_df_d", "I_df_u", "F"),  # _solve_zq_pa.py:413-418
}
_MID_KEYS = {"n79": ("aI_lsl", "aI_lsh")}  # (ncol, nz-1, nb)


def _f64(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.dtype != torch.float64:
        raise TypeError(f"{name} must be float64, got {t.dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {t.device}); crt1d_amd has no CPU path")
    return t.contiguous()


def _fio(t, name):
    """Spectra may be float64 (crt_hip_*_f64) or float32 (crt_hip_*_f32: half the HBM bytes, fp64 arithmetic)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.dtype not in (torch.float64, torch.float32):
        raise TypeError(f"{name} must be float64 or float32, got {t.dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {t.device}); crt1d_amd has no CPU path")
    return t.contiguous()


def _check_profile(t, name, shape, device, dtype=torch.float64):
    """A caller-supplied profile / output array: the C ABI receives a bare pointer, so everything the kernels assume about it
    (dtype, shape, device, contiguity) is checked here."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_cuda or t.device != device:
        raise ValueError(f"{name} must live on {device} (got {t.device})")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


@dataclass
class Columns:
    """Device-resident per-column canopy geometry (what one reference ``Model`` holds)."""

    psi: torch.Tensor  # (ncol,)
    lai: torch.Tensor  # (ncol, nz), index 0 = ground
    g_kind: torch.Tensor  # (ncol,) int32, crt1d_amd.leaf_angle kind ids
    g_param: torch.Tensor  # (ncol,)
    mla: Optional[torch.Tensor] = None  # (ncol,) degrees; 2s only
    g_at_psi: Optional[torch.Tensor] = None  # (ncol,)  G_TABLE columns
    g_table: Optional[torch.Tensor] = None  # (ncol, NQ) G_TABLE columns

    def __post_init__(self):
        self.psi = _f64(self.psi, "psi")
        self.lai = _f64(self.lai, "lai")
        if self.lai.ndim != 2 or self.psi.shape != (self.lai.shape[0],):
            raise ValueError("lai must be (ncol, nz) and psi (ncol,)")
        self.g_param = _f64(self.g_param, "g_param")
        if self.g_kind.dtype != torch.int32 or not self.g_kind.is_cuda:
            raise TypeError("g_kind must be a CUDA int32 tensor")
        self.g_kind = self.g_kind.contiguous()
        for name in ("mla", "g_at_psi"):
            v = getattr(self, name)
            if v is not None:
                v = _f64(v, name)
                if v.shape != self.psi.shape:
                    raise ValueError(f"{name} must be (ncol,)")
                setattr(self, name, v)
        if self.g_table is not None:
            self.g_table = _f64(self.g_table, "g_table")
            if self.g_table.shape != (self.ncol, _lib.NQ):
                raise ValueError(f"g_table must be (ncol, {_lib.NQ})")

    @property
    def ncol(self):
        return self.lai.shape[0]

    @property
    def nz(self):
        return self.lai.shape[1]

    @property
    def device(self):
        return self.lai.device

    def validate(self):
        """The orientation checks ``Model._check_inputs`` makes per column (``crt1d/model.py:244-246``), for the whole
        batch (one device->host sync): lai strictly decreasing with level, lai[:, -1] == 0, 0 <= psi < pi/2."""
        import math

        lai = self.lai
        ok = bool((lai[:, :-1] > lai[:, 1:]).all()) and bool((lai[:, -1] == 0).all())
        if not ok:
            raise AssertionError("lai must decrease strictly from index 0 (ground, total LAI) to 0 at the canopy top")
        if not bool(((self.psi >= 0) & (self.psi < math.pi / 2)).all()):
            raise AssertionError("psi must be in [0, pi/2)")
        if not bool(((self.g_kind >= 0) & (self.g_kind <= 6)).all()):
            raise ValueError("invalid leaf-angle kind")
        self.check_tables()
        return self

    def check_tables(self):
        """Columns of kind G_TABLE dereference ``g_table`` / ``g_at_psi`` on the device: when either is missing make sure no
        column asks for it (one device->host sync, and only in the case where a table is missing)."""
        if getattr(self, "_tables_ok", False):
            return
        if (self.g_table is None or self.g_at_psi is None) and bool((self.g_kind == 6).any()):
            raise ValueError("columns with g_kind = G_TABLE need g_table and g_at_psi")
        self._tables_ok = True  # checked once per object; slices inherit the result (no sync inside a tiled, overlapped loop)

    def slice(self, lo, hi):
        """Columns [lo, hi) as a view (used by the column-sharded multi-GPU path)."""
        g = lambda t: None if t is None else t[lo:hi]  # noqa: E731
        c = Columns(self.psi[lo:hi], self.lai[lo:hi], self.g_kind[lo:hi], self.g_param[lo:hi], g(self.mla),
                    g(self.g_at_psi), g(self.g_table))
        c._tables_ok = getattr(self, "_tables_ok", False)
        return c

    def c_struct(self):
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        return _lib.CrtColumns(self.ncol, self.nz, p(self.psi), p(self.lai), p(self.mla), p(self.g_kind), p(self.g_param),
                               p(self.g_at_psi), p(self.g_table))

    @classmethod
    def from_host(cls, d, device="cuda"):
        """From a dict of NumPy arrays (e.g. :func:`crt1d_amd.synth.make_columns`)."""
        t = lambda k: None if d.get(k) is None else torch.as_tensor(d[k]).to(device)  # noqa: E731
        return cls(t("psi"), t("lai"), t("g_kind"), t("g_param"), t("mla"), t("g_at_psi"), t("g_table"))


@dataclass
class Bands:
    """Per-(column, band) spectra; each tensor is (ncol, nb), or (nb,)/(1, nb) to share one
    spectrum over all columns (the reference's (n_wl,) plugin inputs)."""

    I_dr0: torch.Tensor
    I_df0: torch.Tensor
    leaf_r: torch.Tensor
    leaf_t: torch.Tensor
    soil_r: Optional[torch.Tensor] = None

    def __post_init__(self):
        shape = None
        self.dtype = None
        for name in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r"):
            v = getattr(self, name)
            if v is None:
                continue
            v = _fio(v, name)
            if self.dtype is None:
                self.dtype = v.dtype
            elif v.dtype != self.dtype:
                raise TypeError("all band arrays must share one dtype")
            if v.ndim == 1:
                v = v[None, :]
            if v.ndim != 2:
                raise ValueError(f"{name} must be (ncol, nb) or (nb,)")
            if shape is None:
                shape = v.shape
            elif v.shape != shape:
                raise ValueError("all band arrays must have the same shape")
            setattr(self, name, v)
        self._shape = shape

    @property
    def nb(self):
        return self._shape[1]

    def col_stride(self, ncol):
        if self._shape[0] == 1 and ncol != 1:
            return 0
        if self._shape[0] != ncol:
            raise ValueError(f"band arrays have {self._shape[0]} rows but there are {ncol} columns")
        return self.nb

    def slice(self, lo, hi):
        if self._shape[0] == 1:
            return self
        g = lambda t: None if t is None else t[lo:hi]  # noqa: E731
        return Bands(self.I_dr0[lo:hi], self.I_df0[lo:hi], self.leaf_r[lo:hi], self.leaf_t[lo:hi], g(self.soil_r))

    def band_slice(self, lo, hi):
        """Bands [lo, hi) of every column (band-sharded multi-GPU path); copies to keep rows contiguous."""
        g = lambda t: None if t is None else t[:, lo:hi].contiguous()  # noqa: E731
        return Bands(g(self.I_dr0), g(self.I_df0), g(self.leaf_r), g(self.leaf_t), g(self.soil_r))

    def c_struct(self, ncol):
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        return _lib.CrtBands(self.nb, self.col_stride(ncol), p(self.I_dr0), p(self.I_df0), p(self.leaf_r), p(self.leaf_t),
                             p(self.soil_r))

    @classmethod
    def from_host(cls, d, device="cuda"):
        t = lambda k: None if d.get(k) is None else torch.as_tensor(d[k]).to(device)  # noqa: E731
        return cls(t("I_dr0"), t("I_df0"), t("leaf_r"), t("leaf_t"), t("soil_r"))


def _check_band_device(bands, device):
    for name in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r"):
        v = getattr(bands, name)
        if v is not None and v.device != device:
            raise ValueError(f"{name} lives on {v.device} but the columns on {device}")


def _check_workspace(workspace, need, device):
    if workspace is None:
        return torch.empty(need, dtype=torch.uint8, device=device)
    if not isinstance(workspace, torch.Tensor) or not workspace.is_cuda or workspace.device != device:
        raise ValueError(f"workspace must be a tensor on {device}")
    if not workspace.is_contiguous() or workspace.numel() * workspace.element_size() < need:
        raise ValueError(f"workspace too small or not contiguous (need {need} bytes)")
    return workspace


def workspace_bytes(scheme, ncol, nz, nb=1):
    """Device workspace a solve needs; ``nb`` matters for zq_pa only (its computational-grid fluxes live there)."""
    return int(_lib.load().crt_hip_workspace_bytes_nb(_lib.SCHEME_IDS[scheme], ncol, nz, nb))


class _DeviceBuffer:
    """Owner of one buffer of a ``crt_hip_buffer_alloc_set`` allocation; exposes it to torch through
    ``__cuda_array_interface__`` (the tensor made from it keeps this object, and with it the memory, alive)."""

    def __init__(self, ptr, shape, dtype, device):
        self._lib = _lib.load()
        self._ptr = ctypes.c_void_p(ptr)
        self._dev = torch.device(device)
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape), "typestr": {torch.float64: "<f8", torch.float32: "<f4"}[dtype],
            "data": (ptr, False), "version": 2, "strides": None,
        }

    def classes(self):
        buf = ctypes.create_string_buffer(4096)
        _lib.check(self._lib.crt_hip_buffer_describe(self._ptr, buf, len(buf)), "crt_hip_buffer_describe")
        return buf.value.decode()

    def __del__(self):
        if getattr(self, "_ptr", None) is not None and self._ptr.value:
            try:
                torch.cuda.synchronize(self._dev)  # nothing may still be writing into it
                self._lib.crt_hip_buffer_free(self._ptr)
            except Exception:  # interpreter shutdown
                pass
            self._ptr = None


_OWNERS = "_crt_owner"  # attribute under which a tensor made by device_buffers() carries its _DeviceBuffer


def device_buffers(shapes, dtype=torch.float64, device="cuda"):
    """Tensors for ONE output set (arrays a kernel writes in step), placed by ``crt_hip_buffer_alloc_set``: 512 MB physical chunks
    whose memory classes are interleaved across the arrays (include/crt1d_hip.h; csrc/buffers.hip).  Raises if the HIP
    virtual-memory API is unavailable -- callers that can live with any placement catch the error and use ``torch.empty``."""
    lib = _lib.load()
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    itemsize = torch.empty((), dtype=dtype).element_size()
    n = len(shapes)
    sizes = []
    for shape in shapes:
        k = itemsize
        for s in shape:
            k *= int(s)
        sizes.append(max(k, itemsize))
    c_sizes = (ctypes.c_size_t * n)(*sizes)
    c_ptrs = (ctypes.c_void_p * n)()
    with torch.cuda.device(dev):
        _lib.check(lib.crt_hip_buffer_alloc_set(n, c_sizes, c_ptrs), "crt_hip_buffer_alloc_set")
    out = []
    for ptr, shape in zip(c_ptrs, shapes):
        owner = _DeviceBuffer(ptr, shape, dtype, dev)
        t = torch.as_tensor(owner, device=dev)
        setattr(t, _OWNERS, owner)
        out.append(t)
    return out


def device_buffer(shape, dtype=torch.float64, device="cuda"):
    """One tensor from ``crt_hip_buffer_alloc`` (a set of one)."""
    return device_buffers([shape], dtype, device)[0]


def buffer_classes(t):
    """Memory-class letters of the chunks behind a tensor made by :func:`device_buffers` (``None`` for any other tensor)."""
    owner = getattr(t, _OWNERS, None)
    return None if owner is None else owner.classes()


def buffer_stats(device=None):
    """``crt_hip_buffer_stats`` of the current (or given) device as a dict."""
    lib = _lib.load()
    a = (ctypes.c_int64 * 6)()
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        _lib.check(lib.crt_hip_buffer_stats(a), "crt_hip_buffer_stats")
    return dict(zip(("chunks_created", "chunks_released", "probes", "probe_us", "free_chunks", "classes_seen"), [int(v) for v in a]))


def trim_buffers(device=None):
    """Hand the set allocator's pooled chunks of the current (or given) device back to the driver (``crt_hip_buffer_trim``).  The pool
    keeps at most its retention cap anyway (8 GB by default, :func:`set_pool_retention`); call this before a large allocation through
    another allocator (``torch.empty``, RCCL buffers) when every GB counts."""
    lib = _lib.load()
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        _lib.check(lib.crt_hip_buffer_trim(), "crt_hip_buffer_trim")


def set_pool_retention(nbytes):
    """Cap on the memory the set allocator's per-device pools keep after buffers are freed (``crt_hip_buffer_set_retain``;
    default 8 GB, or ``CRT1D_POOL_RETAIN_MB``).  Applied at once to the existing pools."""
    _lib.check(_lib.load().crt_hip_buffer_set_retain(int(nbytes)), "crt_hip_buffer_set_retain")


PLACED_MIN_BYTES = 1 << 30  # below this the whole output set lives in the 256 MB Infinity Cache / a few chunks: plain torch memory


def alloc_outputs(scheme, ncol, nz, nb, device, dtype=torch.float64, placed=False):
    """Output arrays of one scheme.  ``placed=True``: through :func:`device_buffers` (class-interleaved 512 MB chunks) when the
    set is at least ``PLACED_MIN_BYTES``; otherwise (or if the virtual-memory API fails) plain ``torch.empty``."""
    shapes = {}
    for k in OUT_KEYS[scheme]:
        n = nz - 1 if k in _MID_KEYS.get(scheme, ()) else nz
        shapes[k] = (ncol, n, nb)
    item = 8 if dtype == torch.float64 else 4
    total = sum(s[0] * s[1] * s[2] * item for s in shapes.values())
    if placed and total >= PLACED_MIN_BYTES:
        try:
            return dict(zip(shapes, device_buffers(list(shapes.values()), dtype, device)))
        except RuntimeError:
            pass  # placement is a performance feature: any device memory is correct
    try:
        return {k: torch.empty(s, dtype=dtype, device=device) for k, s in shapes.items()}
    except torch.OutOfMemoryError:
        # what the set allocator's pool still holds is invisible to torch's caching allocator: release it and try once more
        trim_buffers(device)
        torch.cuda.empty_cache()
        return {k: torch.empty(s, dtype=dtype, device=device) for k, s in shapes.items()}


class Plan:
    """Pre-validated launch of one scheme on fixed buffers: ``plan()`` enqueues K0 + the solve kernel
    on the current stream with no allocation and no host synchronisation (bench / steady-state use)."""

    def __init__(self, scheme, cols: Columns, bands: Bands, *, mu_s=0.501, tau_d_method="quad", out=None, workspace=None,
                 placement="auto", tune=None):
        """``placement``: ``"auto"`` allocates output sets of 1 GB and more through the class-interleaving allocator
        (``crt_hip_buffer_alloc_set``: deterministic ~7 TB/s store mode); ``"none"`` uses ``torch.empty``
        (the store rate then depends on where the driver happens to put the arrays).  Ignored when ``out`` is given."""
        if scheme not in _lib.SCHEME_IDS:
            raise ValueError(f"unknown scheme {scheme!r}; valid: {', '.join(SCHEMES)}")
        if tau_d_method not in _lib.TAU_D_METHODS:
            raise ValueError("invalid `method`. Valid options are 'quad' and '9sky'.")  # common.py:78
        self.lib = _lib.load()
        self.scheme = scheme
        self.cols, self.bands = cols, bands
        ncol, nz, nb = cols.ncol, cols.nz, bands.nb
        if scheme == "2s" and cols.mla is None:
            raise ValueError("solve_2s needs `mla`")
        if scheme != "bl" and bands.soil_r is None:
            raise ValueError(f"solve_{scheme} needs `soil_r`")
        cols.check_tables()
        _check_band_device(bands, cols.device)
        if placement not in ("auto", "none"):
            raise ValueError("placement must be 'auto' or 'none'")
        placed = placement == "auto" and os.environ.get("CRT1D_PLACEMENT", "auto") != "none"
        self.out = alloc_outputs(scheme, ncol, nz, nb, cols.device, bands.dtype, placed=placed) if out is None else out
        for k in OUT_KEYS[scheme]:
            if k not in self.out:
                raise ValueError(f"`out` lacks {k!r}")
            rows = nz - 1 if k in _MID_KEYS.get(scheme, ()) else nz
            _check_profile(self.out[k], f"output {k!r}", (ncol, rows, nb), cols.device, bands.dtype)
        need = workspace_bytes(scheme, ncol, nz, nb)
        workspace = _check_workspace(workspace, need, cols.device)
        self.workspace = workspace
        self._c = cols.c_struct()
        self._b = bands.c_struct(ncol)
        self._o = _lib.CrtOptions(float(mu_s), _lib.TAU_D_METHODS[tau_d_method], 0)
        self.set_tune(tune or {})
        if bands.dtype == torch.float32 and scheme not in _lib.F32_SCHEMES:
            raise TypeError(f"scheme {scheme!r} has no f32 storage variant yet")
        self._entry = f"crt_hip_{scheme}_{'f32' if bands.dtype == torch.float32 else 'f64'}"
        self._fn = getattr(self.lib, self._entry)
        self._wsb = workspace.numel() * workspace.element_size()
        self._point_at(self.out)
        # where the output arrays live: memory-class letters per 512 MB chunk for arrays from the set allocator (None = torch memory)
        cls = {k: buffer_classes(v) for k, v in self.out.items()}
        self.placement_report = {"allocator": "crt_hip_buffer_alloc_set", "classes": cls} if any(cls.values()) else None

    def set_tune(self, tune):
        """Measurement aid: per-plan overrides of the kernel-selection heuristics (``crt_options.tune``; keys in
        csrc/crt_internal.hpp).  ``{}`` = automatic.  They travel with every call of this plan -- no process-global state."""
        for k in range(_lib.NTUNE):
            self._o.tune[k] = int(tune.get(k, 0))
        return self

    def last_kernel(self):
        """Name / configuration of the solve kernel this thread's most recent call launched (``crt_hip_last_kernel``)."""
        return self.lib.crt_hip_last_kernel().decode()

    def _point_at(self, out):
        self.out = out
        ptrs = [out[k].data_ptr() for k in OUT_KEYS[self.scheme]]
        ptrs += [None] * (7 - len(ptrs))
        self._out = _lib.CrtOutputs(*ptrs)

    def __call__(self, stream=None, *, flags=0):
        """Enqueue on ``stream`` (default: torch's current stream).  ``flags``: ``_lib.FLAG_SKIP_PRECOMPUTE`` reuses
        the column records already in the workspace (same geometry, new spectra); ``_lib.FLAG_PRECOMPUTE_ONLY``
        runs only the column precompute."""
        dev = self.cols.device
        s = torch.cuda.current_stream(dev) if stream is None else stream
        self._o.flags = int(flags)
        with torch.cuda.device(dev):  # the launch goes to the CURRENT device: make it the one the buffers live on
            st = self._fn(ctypes.byref(self._c), ctypes.byref(self._b), ctypes.byref(self._o), ctypes.byref(self._out),
                          self.workspace.data_ptr(), self._wsb, s.cuda_stream)
        _lib.check(st, self._entry)
        return self.out


def solve(scheme, cols: Columns, bands: Bands, *, mu_s=0.501, tau_d_method="quad", out=None, workspace=None, placement="none"):
    """Run ``scheme`` over all (column, band) pairs; returns a dict of ``(ncol, nz, nb)`` CUDA tensors.

    Asynchronous on the current stream, like any torch op.  One-shot calls take their outputs from torch's caching allocator
    (``placement="none"``): allocation then costs microseconds, but output sets of a GB and more land wherever the driver put them and the
    solve kernel runs 10-20 % below what a class-interleaved set allows (0.77-0.83 of the HBM peak instead of 0.87-0.89, DESIGN.md
    section 3.1).  Steady-state users build a :class:`Plan` once (``placement="auto"``: ~2 ms per 512 MB chunk at construction, nothing
    per call) or pass ``placement="auto"`` here when the set is large enough for that to pay.
    """
    with torch.cuda.device(cols.device):
        return Plan(scheme, cols, bands, mu_s=mu_s, tau_d_method=tau_d_method, out=out, workspace=workspace, placement=placement)()


def _check_epilogue_inputs(cols, bands, sol):
    """The epilogue entry points are fp64 only (``crt_hip_absorb*_f64``): float32 profiles (the output of an f32 solve) read as
    double would run past the end of the allocation."""
    if bands.dtype != torch.float64:
        raise TypeError("the epilogue kernels take float64 spectra and profiles (got float32 bands); upcast first")
    _check_band_device(bands, cols.device)
    bands.col_stride(cols.ncol)
    cols.check_tables()
    for k in ("I_dr", "I_df_d", "I_df_u"):
        if k not in sol:
            raise ValueError(f"`sol` lacks {k!r}")
        _check_profile(sol[k], f"sol[{k!r}]", (cols.ncol, cols.nz, bands.nb), cols.device)


BANDSUM_KEYS = ("aI", "aI_sl", "aI_sh", "totals")
# with ``profiles=True``: the direct-beam part of the absorption and the band-integrated LEVEL profiles of every irradiance variable
# ``diagnostics.band`` reduces (crt1d/diagnostics.py:84-91; the "I..." / "F" variables of ``Model.to_xr``, model.py:421-426)
PROFILE_KEYS = ("aI_dr", "I_dr", "I_df_d", "I_df_u", "F", "I_d")


def bandsum_shapes(ncol, nz, ngroup, profiles=False):
    """Shapes of the integrated outputs, in ``BANDSUM_KEYS`` (+ ``PROFILE_KEYS``) order."""
    sh = {"aI": (ncol, nz - 1, ngroup), "aI_sl": (ncol, nz - 1, ngroup), "aI_sh": (ncol, nz - 1, ngroup), "totals": (ncol, ngroup, 4)}
    if profiles:
        sh["aI_dr"] = (ncol, nz - 1, ngroup)
        for k in PROFILE_KEYS[1:]:
            sh[k] = (ncol, nz, ngroup)
    return sh


def _bandsum_out_struct(out, profiles):
    keys = BANDSUM_KEYS + (PROFILE_KEYS if profiles else ())
    return _lib.CrtBandsumOut(**{k: out[k].data_ptr() for k in keys})


def absorption_from_bandsums(res):
    """The seven entries of the reference's absorption dict (``model.py:637-647``), band-integrated, from a ``profiles=True`` result:
    ``aI_df = aI - aI_dr``, ``aI_df_sl = aI_sl - aI_dr``, ``aI_df_sh = aI_sh`` (``model.py:628-634``)."""
    return {"aI": res["aI"], "aI_dr": res["aI_dr"], "aI_df": res["aI"] - res["aI_dr"], "aI_sl": res["aI_sl"], "aI_sh": res["aI_sh"],
            "aI_df_sl": res["aI_sl"] - res["aI_dr"], "aI_df_sh": res["aI_sh"]}


class BandSumPlan:
    """Pre-validated launch of the epilogue (``crt_hip_absorb_bandsum_f64``) on fixed buffers: ``plan()`` enqueues the kernel with
    no allocation and no host synchronisation.  ``out`` may hold caller-owned output tensors (e.g. views into one packed
    message buffer, :class:`crt1d_amd.dist.BandShardPlan`)."""

    def __init__(self, cols: Columns, bands: Bands, sol, band_w, out=None, profiles=False):
        self.lib = _lib.load()
        band_w = _f64(band_w, "band_w")
        if band_w.ndim == 1:
            band_w = band_w[None, :]
        ng = band_w.shape[0]
        ncol, nz, dev = cols.ncol, cols.nz, cols.device
        _check_epilogue_inputs(cols, bands, sol)
        if band_w.shape[1] != bands.nb or not 1 <= ng <= 4 or band_w.device != dev:
            raise ValueError(f"band_w must be (ngroup <= 4, nb = {bands.nb}) on {dev}")
        shapes = bandsum_shapes(ncol, nz, ng, profiles)
        if out is None:
            out = {k: torch.empty(sh, dtype=torch.float64, device=dev) for k, sh in shapes.items()}
        else:
            for k, sh in shapes.items():
                _check_profile(out[k], f"out[{k!r}]", sh, dev)
        self.cols, self.bands, self.sol, self.band_w, self.out, self.ng, self.profiles = cols, bands, sol, band_w, out, ng, profiles
        self._c, self._b = cols.c_struct(), bands.c_struct(ncol)
        self._o = _bandsum_out_struct(out, profiles)

    def __call__(self, stream=None):
        dev = self.cols.device
        s = torch.cuda.current_stream(dev) if stream is None else stream
        sol = self.sol
        with torch.cuda.device(dev):
            st = self.lib.crt_hip_absorb_bandsum2_f64(
                ctypes.byref(self._c), ctypes.byref(self._b), sol["I_dr"].data_ptr(), sol["I_df_d"].data_ptr(), sol["I_df_u"].data_ptr(),
                self.band_w.data_ptr(), self.ng, ctypes.byref(self._o), s.cuda_stream)
        _lib.check(st, "crt_hip_absorb_bandsum2_f64")
        return self.out


def absorb_bandsum(cols: Columns, bands: Bands, sol, band_w, out=None, profiles=False):
    """Layer absorption (``model.py:573-647``) reduced over bands with weights ``band_w (ngroup, nb)``
    (``diagnostics.py:39-108``).  Returns ``aI, aI_sl, aI_sh`` ``(ncol, nz-1, ngroup)`` and the
    energy-balance terms ``totals (ncol, ngroup, 4)`` = incoming, reflected, transmitted, soil-reflected.
    ``profiles=True`` adds everything else ``diagnostics.band`` returns (``PROFILE_KEYS``): the band-integrated level profiles
    ``I_dr, I_df_d, I_df_u, F, I_d (ncol, nz, ngroup)`` and ``aI_dr (ncol, nz-1, ngroup)`` (:func:`absorption_from_bandsums`).
    Photon-flux variants (``calc_PFD``) are a choice of weights: ``spectra.band_weights(..., wl=..., pfd=True)``."""
    return dict(BandSumPlan(cols, bands, sol, band_w, out=out, profiles=profiles)())


ABSORPTION_KEYS = ("aI", "aI_df", "aI_dr", "aI_sh", "aI_sl", "aI_df_sl", "aI_df_sh")  # model.py:637-647


def absorb(cols: Columns, bands: Bands, sol):
    """Per-band layerwise absorption: the reference's ``Model.absorption`` dict (``model.py:573-647``), batched.
    Returns the seven ``(ncol, nz-1, nb)`` arrays plus ``laim``, ``f_slm`` ``(ncol, nz-1)``."""
    lib = _lib.load()
    ncol, nz, nb = cols.ncol, cols.nz, bands.nb
    dev = cols.device
    _check_epilogue_inputs(cols, bands, sol)
    out = {k: torch.empty((ncol, nz - 1, nb), dtype=torch.float64, device=dev) for k in ABSORPTION_KEYS}
    laim = torch.empty((ncol, nz - 1), dtype=torch.float64, device=dev)
    f_slm = torch.empty_like(laim)
    ptrs = (ctypes.c_void_p * 7)(*[out[k].data_ptr() for k in ABSORPTION_KEYS])
    c, b = cols.c_struct(), bands.c_struct(ncol)
    with torch.cuda.device(dev):
        st = lib.crt_hip_absorb_f64(ctypes.byref(c), ctypes.byref(b), sol["I_dr"].data_ptr(), sol["I_df_d"].data_ptr(),
                                    sol["I_df_u"].data_ptr(), ptrs, laim.data_ptr(), f_slm.data_ptr(),
                                    torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(st, "crt_hip_absorb_f64")
    out["laim"] = laim
    out["f_slm"] = f_slm
    return out


class IntegratedPlan:
    """Fused solve + absorption + band integrals (``crt_hip_integrated_f64``): no profile ever reaches HBM.
    Outputs as :func:`absorb_bandsum`: ``aI, aI_sl, aI_sh (ncol, nz-1, ngroup)``, ``totals (ncol, ngroup, 4)``."""

    def __init__(self, scheme, cols: Columns, bands: Bands, band_w, *, mu_s=0.501, tau_d_method="quad", workspace=None, out=None,
                 profiles=False):
        if scheme not in _lib.SCHEME_IDS or scheme == "zq_pa":
            raise ValueError(f"scheme {scheme!r} has no integrated kernel")
        if tau_d_method not in _lib.TAU_D_METHODS:
            raise ValueError("invalid `method`. Valid options are 'quad' and '9sky'.")
        if bands.dtype != torch.float64:
            raise TypeError("the integrated path takes float64 spectra")
        self.lib = _lib.load()
        self.scheme, self.cols, self.bands = scheme, cols, bands
        band_w = _f64(band_w, "band_w")
        if band_w.ndim == 1:
            band_w = band_w[None, :]
        if band_w.shape[1] != bands.nb or not 1 <= band_w.shape[0] <= 4:
            raise ValueError("band_w must be (ngroup <= 4, nb)")
        self.band_w = band_w
        ncol, nz, ng, dev = cols.ncol, cols.nz, band_w.shape[0], cols.device
        shapes = bandsum_shapes(ncol, nz, ng, profiles)
        if out is None:
            out = {k: torch.empty(sh, dtype=torch.float64, device=dev) for k, sh in shapes.items()}
        else:
            for k, sh in shapes.items():
                _check_profile(out[k], f"out[{k!r}]", sh, dev)
        self.out = out
        self.profiles = profiles
        self._out = _bandsum_out_struct(out, profiles)
        cols.check_tables()
        _check_band_device(bands, dev)
        if band_w.device != dev:
            raise ValueError(f"band_w lives on {band_w.device} but the columns on {dev}")
        if scheme == "2s" and cols.mla is None:
            raise ValueError("solve_2s needs `mla`")
        if scheme != "bl" and bands.soil_r is None:
            raise ValueError(f"solve_{scheme} needs `soil_r`")
        need = workspace_bytes(scheme, ncol, nz, bands.nb)
        self.workspace = _check_workspace(workspace, need, dev)
        self._c, self._b = cols.c_struct(), bands.c_struct(ncol)
        self._o = _lib.CrtOptions(float(mu_s), _lib.TAU_D_METHODS[tau_d_method], 0)

    def __call__(self, stream=None, *, flags=0):
        dev = self.cols.device
        s = torch.cuda.current_stream(dev) if stream is None else stream
        self._o.flags = int(flags)
        with torch.cuda.device(dev):
            st = self.lib.crt_hip_integrated2_f64(
                _lib.SCHEME_IDS[self.scheme], ctypes.byref(self._c), ctypes.byref(self._b), ctypes.byref(self._o), self.band_w.data_ptr(),
                self.band_w.shape[0], ctypes.byref(self._out), self.workspace.data_ptr(),
                self.workspace.numel() * self.workspace.element_size(), s.cuda_stream)
        _lib.check(st, "crt_hip_integrated2_f64")
        return self.out


def solve_integrated(scheme, cols: Columns, bands: Bands, band_w, **kw):
    with torch.cuda.device(cols.device):
        return IntegratedPlan(scheme, cols, bands, band_w, **kw)()
