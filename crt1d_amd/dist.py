"""
Multi-GPU: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm).

The (column x band) solves are independent units; nothing is exchanged inside a solve.

* ``partition="column"`` (preferred): rank r owns a contiguous block of columns, all bands.  Zero
  communication -- integrated quantities are per column and stay on the owning rank.
* ``partition="band"``: rank r owns a contiguous block of bands of ALL columns (300 bands on 8 ranks ->
  38,38,38,38,37,37,37,37).  The per-column precompute is replicated.  The only cross-band coupling is the
  spectral integral ``sum_wl w X`` (``crt1d/diagnostics.py:81``), so each rank forms partial band sums of the
  layer absorption and of the energy-balance terms and ONE all-reduce(sum) of one packed fp64 buffer
  completes them (all band groups in a single call; xGMI rings are per-link bound, so fewer/larger messages).
  Ratios (reflectance = reflected / incoming) are formed after the reduce.

The compute functions are injectable so that the sharding/reduction logic is testable with ``gloo`` on CPU
(tests pass oracle-backed functions); the defaults are the HIP path.
"""

import torch
import torch.distributed as dist


def block_range(n, rank, world):
    """Contiguous balanced block [lo, hi) of ``n`` items for ``rank`` of ``world`` (first ``n % world`` ranks get one more)."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _default_fns():
    from . import batched

    return batched.solve, batched.absorb_bandsum


def solve_sharded(scheme, cols, bands, band_w, *, partition="column", group=None, solve_fn=None, epilogue_fn=None,
                  keep_profiles=True, integrated_fn=None, column_tiles=1, **opts):
    """Solve this rank's shard and return spectrally integrated results.

    ``cols`` / ``bands``: the FULL problem (objects with ``.ncol``, ``.nb``, ``.slice(lo, hi)``, ``.band_slice(lo, hi)``;
    :class:`crt1d_amd.batched.Columns` / ``Bands`` on the GPU).  ``band_w``: ``(ngroup, nb)`` integration weights.

    Returns a dict with
      ``aI, aI_sl, aI_sh``  ``(ncol_local, nz-1, ngroup)``, ``totals (ncol_local, ngroup, 4)``, ``reflectance (ncol_local, ngroup)``,
      ``columns`` = (lo, hi) of the columns these rows describe, and ``profiles`` = this rank's full (unreduced) solver outputs.
    With ``partition="band"`` every rank ends up with the complete integrated result for all columns.
    ``keep_profiles=False`` uses the fused kernel (``crt_hip_integrated_f64``): no profile is ever written to HBM and
    ``profiles`` is ``None``.
    ``column_tiles > 1`` (band partition): the columns are processed in that many tiles and the all-reduce of tile i is
    issued asynchronously while tile i+1 is being solved (at the HBM roofline the reduce of the
    ~80 MB messages of config 4 is comparable to the solve, so it should hide behind it); ``profiles`` is then a list with one
    entry per tile.
    """
    if solve_fn is None or epilogue_fn is None:
        s, e = _default_fns()
        solve_fn = solve_fn or s
        epilogue_fn = epilogue_fn or e
    if not keep_profiles:
        if integrated_fn is None:
            from . import batched

            integrated_fn = batched.solve_integrated
        fused = integrated_fn
        solve_fn = lambda sch, c, b, **o: None  # noqa: E731
        epilogue_fn = None
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if partition == "column":
        lo, hi = block_range(cols.ncol, rank, world)
        c, b = cols.slice(lo, hi), bands.slice(lo, hi)
        sol = solve_fn(scheme, c, b, **opts)
        res = epilogue_fn(c, b, sol, band_w) if keep_profiles else dict(fused(scheme, c, b, band_w, **opts))
        col_range = (lo, hi)
    elif partition == "band":
        lo, hi = block_range(bands.nb, rank, world)
        b = bands.band_slice(lo, hi)
        bw = band_w[:, lo:hi].contiguous()
        keys = ("aI", "aI_sl", "aI_sh", "totals")
        ntile = max(1, min(int(column_tiles), cols.ncol))
        tiles = []  # (partial results, packed buffer, pending all-reduce)
        sols = []
        for t in range(ntile):
            clo, chi = block_range(cols.ncol, t, ntile)
            ct, bt = (cols, b) if ntile == 1 else (cols.slice(clo, chi), b.slice(clo, chi))
            sol_t = solve_fn(scheme, ct, bt, **opts)
            res_t = epilogue_fn(ct, bt, sol_t, bw) if keep_profiles else dict(fused(scheme, ct, bt, bw, **opts))
            flat, work = None, None
            if world > 1:
                flat = torch.cat([res_t[k].reshape(-1) for k in keys])  # one packed message per tile
                # asynchronous: the next tile's kernels are enqueued while this message is on the wire
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=ntile > 1)
            tiles.append((res_t, flat, work))
            sols.append(sol_t)
        parts = {k: [] for k in keys}
        for res_t, flat, work in tiles:
            if work is not None:
                work.wait()
            off = 0
            for k in keys:
                if flat is not None:
                    n = res_t[k].numel()
                    res_t[k] = flat[off:off + n].view_as(res_t[k])
                    off += n
                parts[k].append(res_t[k])
        res = {k: (v[0] if ntile == 1 else torch.cat(v, dim=0)) for k, v in parts.items()}
        sol = sols[0] if ntile == 1 else sols
        col_range = (0, cols.ncol)
    else:
        raise ValueError("partition must be 'column' or 'band'")
    tot = res["totals"]
    res["reflectance"] = tot[..., 1] / tot[..., 0]  # I_df_u[top] / I_d[top]  (diagnostics.py:510-511)
    res["columns"] = col_range
    res["profiles"] = sol
    return res


def gather_columns(local, ncol, group=None):
    """All-gather per-column integrated results of a column-partitioned run into full ``(ncol, ...)`` tensors."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local
    sizes = [block_range(ncol, r, world)[1] - block_range(ncol, r, world)[0] for r in range(world)]
    m = max(sizes)  # all_gather wants equal shapes: pad the (at most one row) shorter blocks
    padded = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)
