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
        if bool((self.g_kind == 6).any()) and (self.g_table is None or self.g_at_psi is None):
            raise ValueError("columns with g_kind = G_TABLE need g_table and g_at_psi")
        return self

    def slice(self, lo, hi):
        """Columns [lo, hi) as a view (used by the column-sharded multi-GPU path)."""
        g = lambda t: None if t is None else t[lo:hi]  # noqa: E731
        return Columns(self.psi[lo:hi], self.lai[lo:hi], self.g_kind[lo:hi], self.g_param[lo:hi], g(self.mla),
                       g(self.g_at_psi), g(self.g_table))

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


def workspace_bytes(scheme, ncol, nz, nb=1):
    """Device workspace a solve needs; ``nb`` matters for zq_pa only (its computational-grid fluxes live there)."""
    return int(_lib.load().crt_hip_workspace_bytes_nb(_lib.SCHEME_IDS[scheme], ncol, nz, nb))


class _DeviceBuffer:
    """Owner of one ``crt_hip_buffer_alloc`` allocation; exposes it to torch through ``__cuda_array_interface__`` (the
    tensor made from it keeps this object, and with it the memory, alive)."""

    def __init__(self, shape, dtype, device):
        self._lib = _lib.load()
        self._ptr = ctypes.c_void_p()
        self._dev = torch.device(device)
        itemsize = torch.empty((), dtype=dtype).element_size()
        n = 1
        for s in shape:
            n *= int(s)
        with torch.cuda.device(self._dev):
            _lib.check(self._lib.crt_hip_buffer_alloc(max(n, 1) * itemsize, ctypes.byref(self._ptr)), "crt_hip_buffer_alloc")
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape), "typestr": {torch.float64: "<f8", torch.float32: "<f4"}[dtype],
            "data": (self._ptr.value, False), "version": 2, "strides": None,
        }

    def __del__(self):
        if getattr(self, "_ptr", None) is not None and self._ptr.value:
            try:
                torch.cuda.synchronize(self._dev)  # nothing may still be writing into it
                self._lib.crt_hip_buffer_free(self._ptr)
            except Exception:  # interpreter shutdown
                pass
            self._ptr = None


def device_buffer(shape, dtype=torch.float64, device="cuda"):
    """A tensor backed by 1 GB physical chunks (``crt_hip_buffer_alloc``, include/crt1d_hip.h); falls back to nothing: raises
    if the virtual-memory API is unavailable."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return torch.as_tensor(_DeviceBuffer(shape, dtype, dev), device=dev)


def alloc_outputs(scheme, ncol, nz, nb, device, dtype=torch.float64, chunked=False):
    """Output arrays of one scheme; ``chunked=True`` backs arrays of at least 64 MB with 1 GB physical chunks."""
    out = {}
    for k in OUT_KEYS[scheme]:
        n = nz - 1 if k in _MID_KEYS.get(scheme, ()) else nz
        shape = (ncol, n, nb)
        big = ncol * n * nb * (8 if dtype == torch.float64 else 4) >= (64 << 20)
        out[k] = device_buffer(shape, dtype, device) if (chunked and big) else torch.empty(shape, dtype=dtype, device=device)
    return out


class Plan:
    """Pre-validated launch of one scheme on fixed buffers: ``plan()`` enqueues K0 + the solve kernel
    on the current stream with no allocation and no host synchronisation (bench / steady-state use)."""

    def __init__(self, scheme, cols: Columns, bands: Bands, *, mu_s=0.501, tau_d_method="quad", out=None, workspace=None,
                 placement="none"):
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
        self.out = alloc_outputs(scheme, ncol, nz, nb, cols.device, bands.dtype) if out is None else out
        for k in OUT_KEYS[scheme]:
            v = self.out[k]
            if v.dtype != bands.dtype or not v.is_cuda or not v.is_contiguous():
                raise TypeError(f"output {k!r} must be a contiguous CUDA tensor of dtype {bands.dtype}")
        need = workspace_bytes(scheme, ncol, nz, nb)
        if workspace is None:
            workspace = torch.empty(need, dtype=torch.uint8, device=cols.device)
        elif workspace.numel() * workspace.element_size() < need:
            raise ValueError("workspace too small")
        self.workspace = workspace
        self._c = cols.c_struct()
        self._b = bands.c_struct(ncol)
        self._o = _lib.CrtOptions(float(mu_s), _lib.TAU_D_METHODS[tau_d_method], 0)
        if bands.dtype == torch.float32 and scheme not in _lib.F32_SCHEMES:
            raise TypeError(f"scheme {scheme!r} has no f32 storage variant yet")
        self._entry = f"crt_hip_{scheme}_{'f32' if bands.dtype == torch.float32 else 'f64'}"
        self._fn = getattr(self.lib, self._entry)
        self._wsb = workspace.numel() * workspace.element_size()
        self._point_at(self.out)
        self.placement_report = None
        if placement == "auto" and out is None and os.environ.get("CRT1D_PLACEMENT", "auto") != "none":
            self._choose_placement()
        elif placement not in ("auto", "none"):
            raise ValueError("placement must be 'auto' or 'none'")

    def _point_at(self, out):
        self.out = out
        ptrs = [out[k].data_ptr() for k in OUT_KEYS[self.scheme]]
        ptrs += [None] * (7 - len(ptrs))
        self._out = _lib.CrtOutputs(*ptrs)

    def _time_ms(self, reps=3):
        dev = self.cols.device
        st = torch.cuda.current_stream(dev)
        self(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            self(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
        e1.record(st)
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def _choose_placement(self, nsets=4, nmix=12, seed=0x5EED):
        """Pick WHERE the output arrays live.  The solve kernels are HBM-write-bound, and on MI355X the same kernel on the same
        data runs in one of two modes depending on where the driver happened to place the output arrays in HBM: ~0.93 ms or
        ~1.12 ms for 2s at 1e4 x 300 x 60, stable for the life of the allocation, nothing to do with their virtual addresses
        (the slow mode shows twice the DRAM-credit stalls at the L2).  Since a Plan's buffers are
        allocated once and reused, it is worth a fraction of a second: allocate a few candidate sets (each after a random-size
        pad, which moves where the next allocation lands; every other one built from 1 GB physical chunks), time the solve on each,
        try a few mixes of arrays across sets, keep the fastest and free the rest.  Skipped when the candidates would not fit comfortably in free memory."""
        import random

        dev = self.cols.device
        total = sum(v.numel() * v.element_size() for v in self.out.values())
        free, _ = torch.cuda.mem_get_info(dev)
        if total < (256 << 20) or (nsets - 1) * total + (1 << 30) > free // 2:
            return
        with torch.cuda.device(dev):
            rng = random.Random(seed)
            self()  # K0 once: the timings below reuse the column records
            sets, pads = [self.out], []
            ncol, nz, nb = self.cols.ncol, self.cols.nz, self.bands.nb
            for i in range(nsets - 1):
                pads.append(torch.empty(rng.randrange(1, 150) << 21, dtype=torch.uint8, device=dev))
                try:  # every other candidate is backed by 1 GB physical chunks (crt_hip_buffer_alloc): same two modes, a faster best case
                    sets.append(alloc_outputs(self.scheme, ncol, nz, nb, dev, self.bands.dtype, chunked=(i % 2 == 0)))
                except RuntimeError:
                    sets.append(alloc_outputs(self.scheme, ncol, nz, nb, dev, self.bands.dtype))
            # yardstick: the streaming-fill rate of this device, measured on one of the arrays; a candidate whose outputs are
            # written faster than 1.02 x that rate ends the search early; otherwise all candidates (~10 timings, ~40 ms) are tried
            big = max(self.out.values(), key=lambda v: v.numel() * v.element_size())
            nfill = (big.numel() * big.element_size() // 16) * 2  # whole 16-B vectors, counted in doubles
            st = torch.cuda.current_stream(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.lib.crt_hip_probe_fill_f64(big.data_ptr(), nfill, 0.0, st.cuda_stream)
            e0.record(st)
            for _ in range(2):
                self.lib.crt_hip_probe_fill_f64(big.data_ptr(), nfill, 0.0, st.cuda_stream)
            e1.record(st)
            e1.synchronize()
            fill_bytes_per_ms = 2 * nfill * 8 / e0.elapsed_time(e1)
            good_ms = total / (1.02 * fill_bytes_per_ms)  # (the best placements beat the grid-stride fill probe by a few per cent)
            best, tbest, tworst, tried = None, float("inf"), 0.0, 0
            trials = []  # (what, ms) in the order tried
            keys = list(self.out)
            cands = list(sets) + [None] * nmix  # None = a random mix of arrays across the sets
            for ci, cand in enumerate(cands):
                what = "torch" if ci == 0 else ("chunked" if ci % 2 == 1 else "torch+pad") if ci < nsets else "mix"
                if cand is None:
                    cand = {k: sets[rng.randrange(nsets)][k] for k in keys}
                self._point_at(cand)
                t = self._time_ms()
                trials.append((what, round(t, 4)))
                tried += 1
                tworst = max(tworst, t)
                if t < tbest:
                    best, tbest = cand, t
                if tbest <= good_ms:
                    break
            # nothing near the fill rate: every region tried so far is a slow one.  Jump further: behind a multi-GB pad, two fresh
            # sets and a few mixes of their arrays land in other physical regions (bounded by a quarter of the free memory)
            ok_ms = total / (0.93 * fill_bytes_per_ms)
            for pad_gb in (8, 16, 32, 48):
                if tbest <= ok_ms:
                    break
                free, _ = torch.cuda.mem_get_info(dev)
                if (pad_gb << 30) + 2 * total > free // 4:
                    break
                pads.append(torch.empty(pad_gb << 30, dtype=torch.uint8, device=dev))
                far = []
                for chunked in (False, True):
                    try:
                        far.append(alloc_outputs(self.scheme, ncol, nz, nb, dev, self.bands.dtype, chunked=chunked))
                    except RuntimeError:
                        far.append(alloc_outputs(self.scheme, ncol, nz, nb, dev, self.bands.dtype))
                sets.extend(far)
                for j in range(6):
                    cand = far[j] if j < 2 else {k: far[rng.randrange(2)][k] for k in keys}
                    self._point_at(cand)
                    t = self._time_ms()
                    trials.append((f"+{pad_gb}GB {'torch' if j == 0 else 'chunked' if j == 1 else 'mix'}", round(t, 4)))
                    tried += 1
                    tworst = max(tworst, t)
                    if t < tbest:
                        best, tbest = cand, t
                    if tbest <= ok_ms:
                        break
            self._point_at(best)
            self.placement_report = {"candidates_timed": tried, "best_ms": tbest, "worst_ms": tworst, "fill_rate_ms": total / fill_bytes_per_ms,
                                     "trials": trials}
            del sets, pads
            torch.cuda.empty_cache()

    def __call__(self, stream=None, *, flags=0):
        """Enqueue on ``stream`` (default: torch's current stream).  ``flags``: ``_lib.FLAG_SKIP_PRECOMPUTE`` reuses
        the column records already in the workspace (same geometry, new spectra); ``_lib.FLAG_PRECOMPUTE_ONLY``
        runs only the column precompute."""
        s = torch.cuda.current_stream(self.cols.device) if stream is None else stream
        self._o.flags = int(flags)
        st = self._fn(ctypes.byref(self._c), ctypes.byref(self._b), ctypes.byref(self._o), ctypes.byref(self._out),
                      self.workspace.data_ptr(), self._wsb, s.cuda_stream)
        _lib.check(st, self._entry)
        return self.out


def solve(scheme, cols: Columns, bands: Bands, *, mu_s=0.501, tau_d_method="quad", out=None, workspace=None):
    """Run ``scheme`` over all (column, band) pairs; returns a dict of ``(ncol, nz, nb)`` CUDA tensors.

    Asynchronous on the current stream, like any torch op.
    """
    with torch.cuda.device(cols.device):
        return Plan(scheme, cols, bands, mu_s=mu_s, tau_d_method=tau_d_method, out=out, workspace=workspace)()


def absorb_bandsum(cols: Columns, bands: Bands, sol, band_w):
    """Layer absorption (``model.py:573-647``) reduced over bands with weights ``band_w (ngroup, nb)``
    (``diagnostics.py:39-108``).  Returns ``aI, aI_sl, aI_sh`` ``(ncol, nz-1, ngroup)`` and the
    energy-balance terms ``totals (ncol, ngroup, 4)`` = incoming, reflected, transmitted, soil-reflected."""
    lib = _lib.load()
    band_w = _f64(band_w, "band_w")
    if band_w.ndim == 1:
        band_w = band_w[None, :]
    ng = band_w.shape[0]
    ncol, nz = cols.ncol, cols.nz
    dev = cols.device
    aI = torch.empty((ncol, nz - 1, ng), dtype=torch.float64, device=dev)
    aI_sl = torch.empty_like(aI)
    aI_sh = torch.empty_like(aI)
    totals = torch.empty((ncol, ng, 4), dtype=torch.float64, device=dev)
    c, b = cols.c_struct(), bands.c_struct(ncol)
    with torch.cuda.device(dev):
        st = lib.crt_hip_absorb_bandsum_f64(
            ctypes.byref(c), ctypes.byref(b), sol["I_dr"].data_ptr(), sol["I_df_d"].data_ptr(), sol["I_df_u"].data_ptr(),
            band_w.data_ptr(), ng, aI.data_ptr(), aI_sl.data_ptr(), aI_sh.data_ptr(), totals.data_ptr(),
            torch.cuda.current_stream(dev).cuda_stream,
        )
    _lib.check(st, "crt_hip_absorb_bandsum_f64")
    return {"aI": aI, "aI_sl": aI_sl, "aI_sh": aI_sh, "totals": totals}


ABSORPTION_KEYS = ("aI", "aI_df", "aI_dr", "aI_sh", "aI_sl", "aI_df_sl", "aI_df_sh")  # model.py:637-647


def absorb(cols: Columns, bands: Bands, sol):
    """Per-band layerwise absorption: the reference's ``Model.absorption`` dict (``model.py:573-647``), batched.
    Returns the seven ``(ncol, nz-1, nb)`` arrays plus ``laim``, ``f_slm`` ``(ncol, nz-1)``."""
    lib = _lib.load()
    ncol, nz, nb = cols.ncol, cols.nz, bands.nb
    dev = cols.device
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

    def __init__(self, scheme, cols: Columns, bands: Bands, band_w, *, mu_s=0.501, tau_d_method="quad", workspace=None):
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
        self.out = {k: torch.empty((ncol, nz - 1, ng), dtype=torch.float64, device=dev) for k in ("aI", "aI_sl", "aI_sh")}
        self.out["totals"] = torch.empty((ncol, ng, 4), dtype=torch.float64, device=dev)
        need = workspace_bytes(scheme, ncol, nz, bands.nb)
        self.workspace = torch.empty(need, dtype=torch.uint8, device=dev) if workspace is None else workspace
        self._c, self._b = cols.c_struct(), bands.c_struct(ncol)
        self._o = _lib.CrtOptions(float(mu_s), _lib.TAU_D_METHODS[tau_d_method], 0)

    def __call__(self, stream=None, *, flags=0):
        s = torch.cuda.current_stream(self.cols.device) if stream is None else stream
        self._o.flags = int(flags)
        o = self.out
        st = self.lib.crt_hip_integrated_f64(
            _lib.SCHEME_IDS[self.scheme], ctypes.byref(self._c), ctypes.byref(self._b), ctypes.byref(self._o), self.band_w.data_ptr(),
            self.band_w.shape[0], o["aI"].data_ptr(), o["aI_sl"].data_ptr(), o["aI_sh"].data_ptr(), o["totals"].data_ptr(),
            self.workspace.data_ptr(), self.workspace.numel(), s.cuda_stream)
        _lib.check(st, "crt_hip_integrated_f64")
        return self.out


def solve_integrated(scheme, cols: Columns, bands: Bands, band_w, **kw):
    with torch.cuda.device(cols.device):
        return IntegratedPlan(scheme, cols, bands, band_w, **kw)()
