"""
Multi-GPU: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm).

The (column x band) solves are independent units; nothing is exchanged inside a solve.

* ``partition="column"`` (preferred): rank r owns a contiguous block of columns, all bands.  Zero
  communication -- integrated quantities are per column and stay on the owning rank.
* ``partition="band"``: rank r owns a contiguous block of bands of ALL columns (300 bands on 8 ranks ->
  38 x 6 + 36 x 2: dealt in pairs, :func:`band_block_range`).  The per-column precompute is replicated.  The only cross-band coupling is the
  spectral integral ``sum_wl w X`` (``crt1d/diagnostics.py:81``), so each rank forms partial band sums of the
  layer absorption and of the energy-balance terms and ONE all-reduce(sum) of one packed fp64 buffer per column tile
  completes them (all band groups in a single call; xGMI rings are per-link bound, so fewer/larger messages).
  Ratios (reflectance = reflected / incoming, ``diagnostics.py:510-511``) are formed after the reduce.

:class:`BandShardPlan` is the steady-state form of the band partition (buffers and launches planned once; ``bench.py
--partition band`` times it); :func:`solve_sharded` is the one-shot form.  The compute functions are injectable so that the
sharding / packing / reduction logic is testable with ``gloo`` on CPU (tests pass oracle-backed functions); the defaults are
the HIP path.
"""

import torch
import torch.distributed as dist

KEYS = ("aI", "aI_sl", "aI_sh", "totals")
# what travels: aI = aI_sl + aI_sh (model.py:633-635), so the total is NOT sent -- it is re-formed from the reduced sunlit and shaded
# parts after the all-reduce (a third less on the wire; at 8 ranks the ring all-reduce of config 4's messages is comparable to a
# rank's compute).  The re-formed sum differs from a directly reduced aI by rounding only (~1e-16 relative).
MSG_KEYS = ("aI_sl", "aI_sh", "totals")


def block_range(n, rank, world, unit=1):
    """Contiguous balanced block [lo, hi) of ``n`` items for ``rank`` of ``world`` (first ranks get one more).  ``unit > 1``
    hands the items out in groups of ``unit`` (``n`` must be a multiple of it)."""
    if unit > 1:
        lo, hi = block_range(n // unit, rank, world)
        return lo * unit, hi * unit
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def band_block_range(nb, rank, world):
    """This rank's bands.  An even number of bands is dealt in PAIRS, so that every shard is even (300 bands on 8 ranks ->
    38 x 6 + 36 x 2 instead of 38 x 4 + 37 x 4: the largest shard is the same, but the kernels' 16-byte-per-lane fused flush needs
    an even band count -- an odd shard runs the generic flush at 0.75 of the rate, tools/ab_narrow.py -- and the slowest rank sets
    the step time)."""
    return block_range(nb, rank, world, unit=2 if nb % 2 == 0 and nb >= 2 * world else 1)


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _shapes(ncol, nz, ng):
    return {"aI": (ncol, nz - 1, ng), "aI_sl": (ncol, nz - 1, ng), "aI_sh": (ncol, nz - 1, ng), "totals": (ncol, ng, 4)}


class _Tile:
    """One column tile of a band-sharded step: its packed message buffer, views into it, and the launch closure."""

    def __init__(self, clo, chi, nz, ng, like):
        self.clo, self.chi = clo, chi
        shapes = _shapes(chi - clo, nz, ng)

        def numel(k):
            m = 1
            for s in shapes[k]:
                m *= s
            return m

        self.flat = torch.empty(sum(numel(k) for k in MSG_KEYS), dtype=torch.float64, device=like.device)  # the all-reduce message
        self.views, off = {}, 0
        for k in MSG_KEYS:
            self.views[k] = self.flat[off:off + numel(k)].view(shapes[k])
            off += numel(k)
        self.views["aI"] = torch.empty(shapes["aI"], dtype=torch.float64, device=like.device)  # local; re-formed after a reduce
        self.launch = None   # () -> profiles or None; fills self.flat and views["aI"]
        self.work = None
        self.reduced = False
        self.profiles = None


class BandShardPlan:
    """Band-sharded steady-state step (BASELINE.json configs[3]): this rank solves ITS bands of ALL columns, column tile by
    column tile; every tile's partial band sums are written by the epilogue kernel straight into one packed fp64 message and
    all-reduced (RCCL) asynchronously while the next tile is being solved (at the HBM roofline the reduce
    of config 4's messages costs as much as a solve, so it has to hide behind one).

    ``plan()`` enqueues one step and returns after issuing the last collective; ``plan.wait()`` completes the collectives
    (stream-ordered for RCCL: the host does not block) and returns the integrated results for all columns -- complete on
    every rank.  Defaults are the HIP kernels (:class:`crt1d_amd.batched.Plan` + ``BandSumPlan``, or the fused
    ``IntegratedPlan`` with ``keep_profiles=False``); ``solve_fn`` / ``epilogue_fn`` / ``integrated_fn`` replace them
    (functional form, results copied into the message) for the CPU/gloo tests.

    ``share_profiles=True``: all tiles write their profiles into ONE set of output buffers (memory of one tile instead of all
    of them; config 4 on one GPU is 168 GB otherwise).  The profiles of earlier tiles are then gone after the step.
    """

    def __init__(self, scheme, cols, bands, band_w, *, group=None, column_tiles=1, keep_profiles=True, share_profiles=False,
                 solve_fn=None, epilogue_fn=None, integrated_fn=None, placement="auto", always_reduce=False, **opts):
        self.scheme, self.group = scheme, group
        self.always_reduce = always_reduce  # issue the all-reduces even in a world of one rank (API rehearsal)
        self.world, self.rank = _world(group)
        self.ncol, self.nz = cols.ncol, cols.nz
        self.band_range = band_block_range(bands.nb, self.rank, self.world)
        lo, hi = self.band_range
        b = bands.band_slice(lo, hi)
        bw = band_w[:, lo:hi].contiguous()
        self.ng = bw.shape[0]
        self.ntile = max(1, min(int(column_tiles), cols.ncol))
        self.keep_profiles = keep_profiles
        functional = solve_fn is not None or epilogue_fn is not None or integrated_fn is not None
        if functional and keep_profiles and (solve_fn is None or epilogue_fn is None):
            raise ValueError("give both solve_fn and epilogue_fn (or neither)")
        self.tiles = []
        shared = None
        for t in range(self.ntile):
            clo, chi = block_range(cols.ncol, t, self.ntile)
            ct, bt = (cols, b) if self.ntile == 1 else (cols.slice(clo, chi), b.slice(clo, chi))
            tile = _Tile(clo, chi, self.nz, self.ng, bw)
            if functional:
                tile.launch = self._functional_launch(tile, ct, bt, bw, solve_fn, epilogue_fn, integrated_fn, opts)
            else:
                from . import batched

                if keep_profiles:
                    if share_profiles and shared is not None and chi - clo == shared[0]:
                        out = shared[1]
                    else:
                        out = None
                    plan = batched.Plan(scheme, ct, bt, out=out, placement=placement, **opts)
                    if share_profiles and shared is None:
                        shared = (chi - clo, plan.out)
                    epi = batched.BandSumPlan(ct, bt, plan.out, bw, out=tile.views)
                    tile.launch = self._planned_launch(plan, epi)
                    tile.kernel_plan = plan
                else:
                    ip = batched.IntegratedPlan(scheme, ct, bt, bw, out=tile.views, **opts)
                    tile.launch = lambda ip=ip: (ip(), None)[1]
                    tile.kernel_plan = ip
            self.tiles.append(tile)

    @staticmethod
    def _planned_launch(plan, epi):
        def go():
            out = plan()
            epi()
            return out

        return go

    def _functional_launch(self, tile, ct, bt, bw, solve_fn, epilogue_fn, integrated_fn, opts):
        def go():
            if self.keep_profiles:
                sol = solve_fn(self.scheme, ct, bt, **opts)
                res = epilogue_fn(ct, bt, sol, bw)
            else:
                sol = None
                res = dict(integrated_fn(self.scheme, ct, bt, bw, **opts))
            for k in KEYS:
                tile.views[k].copy_(res[k])
            return sol

        return go

    @property
    def message_bytes(self):
        """Bytes all-reduced per step by this rank (sum over the column tiles)."""
        return sum(t.flat.numel() * 8 for t in self.tiles)

    def __call__(self, reduce=True):
        """Enqueue one step.  ``reduce=False`` skips the collectives (compute-only timing).

        A step whose collectives have not been completed (no :meth:`wait` since the last call) is completed first: the epilogue of this
        step writes into the very message buffers (``tile.flat``) the previous step's asynchronous all-reduce may still be reading on the
        RCCL stream."""
        for tile in self.tiles:
            if tile.work is not None:
                tile.work.wait()  # stream-ordered for RCCL: the host does not block, the compute stream waits for the collective
                tile.work = None
            tile.profiles = tile.launch()
            tile.work = None
            tile.reduced = False
            if reduce and (self.world > 1 or (self.always_reduce and dist.is_initialized())):
                # asynchronous: the next tile's kernels are enqueued while this message is on the wire
                tile.work = dist.all_reduce(tile.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                tile.reduced = self.world > 1
        return self

    def wait(self):
        """Complete the step's collectives; returns ``aI, aI_sl, aI_sh (ncol, nz-1, ngroup)``, ``totals (ncol, ngroup, 4)``,
        ``reflectance (ncol, ngroup)`` for ALL columns, plus ``columns`` and ``profiles`` (this rank's bands; a list per tile)."""
        for tile in self.tiles:
            if tile.work is not None:
                tile.work.wait()
                tile.work = None
            if tile.reduced:  # the total from its reduced parts (the local aI held this rank's partial sum only)
                torch.add(tile.views["aI_sl"], tile.views["aI_sh"], out=tile.views["aI"])
                tile.reduced = False
        if self.ntile == 1:
            res = dict(self.tiles[0].views)
            prof = self.tiles[0].profiles
        else:
            res = {k: torch.cat([t.views[k] for t in self.tiles], dim=0) for k in KEYS}
            prof = [t.profiles for t in self.tiles]
        tot = res["totals"]
        res["reflectance"] = tot[..., 1] / tot[..., 0]  # I_df_u[top] / I_d[top]  (diagnostics.py:510-511)
        res["columns"] = (0, self.ncol)
        res["profiles"] = prof if self.keep_profiles else None
        return res


def _default_fns():
    from . import batched

    return batched.solve, batched.absorb_bandsum


def solve_sharded(scheme, cols, bands, band_w, *, partition="column", group=None, solve_fn=None, epilogue_fn=None,
                  keep_profiles=True, integrated_fn=None, column_tiles=1, **opts):
    """Solve this rank's shard once and return spectrally integrated results.

    ``cols`` / ``bands``: the FULL problem (objects with ``.ncol``, ``.nb``, ``.slice(lo, hi)``, ``.band_slice(lo, hi)``;
    :class:`crt1d_amd.batched.Columns` / ``Bands`` on the GPU).  ``band_w``: ``(ngroup, nb)`` integration weights.

    Returns a dict with
      ``aI, aI_sl, aI_sh``  ``(ncol_local, nz-1, ngroup)``, ``totals (ncol_local, ngroup, 4)``, ``reflectance (ncol_local, ngroup)``,
      ``columns`` = (lo, hi) of the columns these rows describe, and ``profiles`` = this rank's full (unreduced) solver outputs.
    With ``partition="band"`` every rank ends up with the complete integrated result for all columns (:class:`BandShardPlan`).
    ``keep_profiles=False`` uses the fused kernel (``crt_hip_integrated_f64``): no profile is ever written to HBM and
    ``profiles`` is ``None``.
    ``column_tiles > 1`` (band partition): the columns are processed in that many tiles and the all-reduce of tile i is
    issued asynchronously while tile i+1 is being solved; ``profiles`` is then a list with one entry per tile.
    """
    if partition == "band":
        if keep_profiles and solve_fn is None and epilogue_fn is None:
            solve_fn, epilogue_fn = _default_fns()  # one-shot: outputs from torch's allocator, nothing planned ahead
        elif keep_profiles and (solve_fn is None or epilogue_fn is None):
            s, e = _default_fns()
            solve_fn, epilogue_fn = solve_fn or s, epilogue_fn or e
        if not keep_profiles and integrated_fn is None:
            from . import batched

            integrated_fn = batched.solve_integrated
        plan = BandShardPlan(scheme, cols, bands, band_w, group=group, column_tiles=column_tiles, keep_profiles=keep_profiles,
                             solve_fn=solve_fn if keep_profiles else None, epilogue_fn=epilogue_fn if keep_profiles else None,
                             integrated_fn=None if keep_profiles else integrated_fn, **opts)
        return plan().wait()
    if partition != "column":
        raise ValueError("partition must be 'column' or 'band'")
    if solve_fn is None or epilogue_fn is None:
        s, e = _default_fns()
        solve_fn = solve_fn or s
        epilogue_fn = epilogue_fn or e
    world, rank = _world(group)
    lo, hi = block_range(cols.ncol, rank, world)
    c, b = cols.slice(lo, hi), bands.slice(lo, hi)
    if keep_profiles:
        sol = solve_fn(scheme, c, b, **opts)
        res = dict(epilogue_fn(c, b, sol, band_w))
    else:
        if integrated_fn is None:
            from . import batched

            integrated_fn = batched.solve_integrated
        sol = None
        res = dict(integrated_fn(scheme, c, b, band_w, **opts))
    tot = res["totals"]
    res["reflectance"] = tot[..., 1] / tot[..., 0]  # I_df_u[top] / I_d[top]  (diagnostics.py:510-511)
    res["columns"] = (lo, hi)
    res["profiles"] = sol
    return res


def grid_mean(res, ncol_total, group=None):
    """Domain means of the integrated results of a COLUMN-partitioned run: every rank sums its own columns' ``aI, aI_sl, aI_sh
    (ncol_local, nz-1, ngroup)`` and ``totals (ncol_local, ngroup, 4)``, ONE small all-reduce (RCCL) adds the ranks' sums, and the
    means and the grid reflectance (reflected / incoming, ``diagnostics.py:510-511``) are formed after it -- "the final reduce of
    spectrally-integrated absorption / reflectance" of the column partition.  Every rank gets the same result."""
    keys = ("aI", "aI_sl", "aI_sh", "totals")
    flat = torch.cat([res[k].sum(dim=0).reshape(-1) for k in keys])
    world, _ = _world(group)
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat = flat / float(ncol_total)
    out, off = {}, 0
    for k in keys:
        shape = res[k].shape[1:]
        n = 1
        for x in shape:
            n *= x
        out[k] = flat[off:off + n].view(shape)
        off += n
    out["reflectance"] = out["totals"][..., 1] / out["totals"][..., 0]
    return out


def gather_columns(local, ncol, group=None):
    """All-gather per-column integrated results of a column-partitioned run into full ``(ncol, ...)`` tensors."""
    world, _ = _world(group)
    if world == 1:
        return local
    sizes = [block_range(ncol, r, world)[1] - block_range(ncol, r, world)[0] for r in range(world)]
    m = max(sizes)  # all_gather wants equal shapes: pad the (at most one row) shorter blocks
    padded = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)
