"""CPU, world_size 2, gloo: the sharding + final-reduce logic of crt1d_amd.dist with an oracle-backed compute
function injected (the HIP kernels need a GPU; the partition / packing / all-reduce / gather code is the same)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

NCOL, NB, NZ = 7, 9, 12  # deliberately not divisible by 2


class HostCols:
    def __init__(self, d):
        self.d = d
        self.ncol, self.nz = d["lai"].shape

    def slice(self, lo, hi):
        return HostCols({k: (v[lo:hi] if isinstance(v, np.ndarray) and v.shape[:1] == (self.ncol,) else v) for k, v in self.d.items()})


class HostBands:
    keys = ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r")

    def __init__(self, d):
        self.d = {k: d[k] for k in self.keys}
        self.nb = d["I_dr0"].shape[1]

    def slice(self, lo, hi):
        return HostBands({k: v[lo:hi] for k, v in self.d.items()})

    def band_slice(self, lo, hi):
        return HostBands({k: np.ascontiguousarray(v[:, lo:hi]) for k, v in self.d.items()})


def _oracle_fns():
    from oracle import crt_oracle as O

    def ocols(c):
        d = c.d
        return O.Columns(d["psi"], d["lai"], mla=d["mla"], g_kind=d["g_kind"], g_param=d["g_param"])

    def solve_fn(scheme, cols, bands, **opts):
        kw = dict(bands.d)
        if scheme == "bl":
            kw.pop("soil_r")
        return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in O.SOLVERS[scheme](ocols(cols), **kw, **opts).items()}

    def epilogue_fn(cols, bands, sol, band_w):
        out = {k: sol[k].numpy() for k in ("I_dr", "I_df_d", "I_df_u")}
        ab = O.calc_absorption(ocols(cols), out, leaf_r=bands.d["leaf_r"], leaf_t=bands.d["leaf_t"])
        w = band_w.numpy().T  # (nb, ng)
        tot = np.stack([(out["I_dr"][:, -1] + out["I_df_d"][:, -1]) @ w, out["I_df_u"][:, -1] @ w,
                        (out["I_dr"][:, 0] + out["I_df_d"][:, 0]) @ w, out["I_df_u"][:, 0] @ w], axis=-1)
        return {"aI": torch.from_numpy(ab["aI"] @ w), "aI_sl": torch.from_numpy(ab["aI_sl"] @ w),
                "aI_sh": torch.from_numpy(ab["aI_sh"] @ w), "totals": torch.from_numpy(tot)}

    return solve_fn, epilogue_fn


def _problem():
    from crt1d_amd import spectra, synth

    d = synth.make_columns(NCOL, NB, NZ, seed=21)
    return d, torch.from_numpy(spectra.band_weights(d["wle"]))


def _worker(rank, world, port, scheme, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from crt1d_amd.dist import gather_columns, solve_sharded

        d, bw = _problem()
        solve_fn, epi = _oracle_fns()
        cols, bands = HostCols(d), HostBands(d)
        rb = solve_sharded(scheme, cols, bands, bw, partition="band", solve_fn=solve_fn, epilogue_fn=epi)
        rc = solve_sharded(scheme, cols, bands, bw, partition="column", solve_fn=solve_fn, epilogue_fn=epi)
        # column tiles with overlapped (asynchronous) all-reduces must give the same integrated results as one message
        rt = solve_sharded(scheme, cols, bands, bw, partition="band", solve_fn=solve_fn, epilogue_fn=epi, column_tiles=3)
        assert isinstance(rt["profiles"], list) and len(rt["profiles"]) == 3
        for k in ("aI", "aI_sl", "aI_sh", "totals", "reflectance"):
            assert rt[k].shape == rb[k].shape
            np.testing.assert_allclose(rt[k].numpy(), rb[k].numpy(), rtol=1e-13, atol=1e-300)
        full_c = {k: gather_columns(rc[k], NCOL) for k in ("aI", "aI_sl", "aI_sh", "totals", "reflectance")}
        q.put((rank, {k: rb[k].numpy() for k in full_c}, {k: v.numpy() for k, v in full_c.items()}, rc["columns"],
               tuple(rb["profiles"]["I_dr"].shape)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("scheme", ["2s", "n79"])
def test_band_and_column_partition_world2(scheme):
    from crt1d_amd.dist import solve_sharded

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, scheme, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference of the same pipeline
    d, bw = _problem()
    solve_fn, epi = _oracle_fns()
    ref = solve_sharded(scheme, HostCols(d), HostBands(d), bw, partition="column", solve_fn=solve_fn, epilogue_fn=epi)
    for rank, band_res, col_res, col_range, prof_shape in got:
        assert col_range == ((0, 4) if rank == 0 else (4, 7))
        assert prof_shape == (NCOL, NZ, 5 if rank == 0 else 4)  # 9 bands -> 5 + 4
        for k in ("aI", "aI_sl", "aI_sh", "totals", "reflectance"):
            # all-reduce order differs from a sequential sum: compare at 1e-12, not bitwise
            np.testing.assert_allclose(band_res[k], ref[k].numpy(), rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(col_res[k], ref[k].numpy(), rtol=1e-14, atol=0)
    # energy: incoming - reflected - soil-absorbed == canopy absorption (solar band group = index 2)
    tot = ref["totals"].numpy()[:, 2]
    canopy = ref["aI"].numpy()[:, :, 2].sum(axis=1)
    np.testing.assert_allclose(tot[:, 0] - tot[:, 1] - (tot[:, 2] - tot[:, 3]), canopy, rtol=1e-10)


def _plan_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from crt1d_amd.dist import BandShardPlan

        d, bw = _problem()
        solve_fn, epi = _oracle_fns()
        plan = BandShardPlan("2s", HostCols(d), HostBands(d), bw, column_tiles=3, solve_fn=solve_fn, epilogue_fn=epi)
        r1 = {k: v.clone() for k, v in plan().wait().items() if k in ("aI", "totals", "reflectance")}
        r2 = plan().wait()  # the same plan again: buffers reused, same answer (steady-state use, bench.py --partition band)
        r2 = {k: r2[k].numpy().copy() for k in r1}
        local = plan(reduce=False).wait()  # compute-only: this rank's partial sums, NOT the full integral
        q.put((rank, plan.band_range, plan.message_bytes, plan.ntile, {k: r1[k].numpy() for k in r1}, r2, float(local["totals"].sum())))
    finally:
        dist.destroy_process_group()


def test_band_shard_plan_is_reusable_world2():
    from crt1d_amd.dist import solve_sharded

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d, bw = _problem()
    solve_fn, epi = _oracle_fns()
    ref = solve_sharded("2s", HostCols(d), HostBands(d), bw, partition="column", solve_fn=solve_fn, epilogue_fn=epi)
    ng = bw.shape[0]
    nbytes = 8 * (2 * NCOL * (NZ - 1) * ng + NCOL * ng * 4)  # aI_sl, aI_sh, totals travel; aI is re-formed after the reduce
    partial = []
    for rank, band_range, msg_bytes, ntile, r1, r2, local_sum in got:
        assert band_range == ((0, 5) if rank == 0 else (5, 9)) and ntile == 3 and msg_bytes == nbytes
        for k in r1:
            np.testing.assert_allclose(r1[k], ref[k].numpy(), rtol=1e-12, atol=1e-13)
            np.testing.assert_array_equal(r1[k], r2[k])
        partial.append(local_sum)
    np.testing.assert_allclose(sum(partial), float(ref["totals"].sum()), rtol=1e-12)


def test_bench_launcher_spawns_ranks_itself():
    """`python bench.py --gpus 2` without torch.distributed.run: the parent starts the two ranks as children (RANK / WORLD_SIZE /
    MASTER_* set, rendezvous on 127.0.0.1) and rank 0 prints the one JSON line.  --launch-check stops before any GPU work."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    assert out == {"launch_check": True, "n_gpus": 2, "ranks_seen": 2, "master": "127.0.0.1"}
    # ranks that fail make the launcher fail (here: an argument the ranks reject)
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check", "--partition", "nonsense"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert bad.returncode != 0


def _gm_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from crt1d_amd.dist import grid_mean, solve_sharded

        d, bw = _problem()
        solve_fn, epi = _oracle_fns()
        rc = solve_sharded("2s", HostCols(d), HostBands(d), bw, partition="column", solve_fn=solve_fn, epilogue_fn=epi)
        g = grid_mean(rc, NCOL)
        q.put((rank, {k: v.numpy().copy() for k, v in g.items()}))
    finally:
        dist.destroy_process_group()


def test_grid_mean_of_the_column_partition_world2():
    """The column partition's final reduce: grid means of the integrated absorption / reflectance over the columns of all ranks."""
    from crt1d_amd.dist import solve_sharded

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d, bw = _problem()
    solve_fn, epi = _oracle_fns()
    ref = solve_sharded("2s", HostCols(d), HostBands(d), bw, partition="column", solve_fn=solve_fn, epilogue_fn=epi)  # world of one
    for _, g in got:
        for k in ("aI", "aI_sl", "aI_sh", "totals"):
            np.testing.assert_allclose(g[k], ref[k].numpy().mean(axis=0), rtol=1e-12, atol=1e-14)
        tot = ref["totals"].numpy().sum(axis=0)
        np.testing.assert_allclose(g["reflectance"], tot[:, 1] / tot[:, 0], rtol=1e-12)
    for k in got[0][1]:
        np.testing.assert_array_equal(got[0][1][k], got[1][1][k])


# ---------------------------------------------------------------------------------------------------------------------
# more ranks than the round-end box can give: the band partition on 3 / 5 / 8 gloo ranks
NB_EVEN = 40


def _problem_even():
    from crt1d_amd import spectra, synth

    d = synth.make_columns(5, NB_EVEN, 10, seed=33, uniform_dlai=False)
    return d, torch.from_numpy(spectra.band_weights(d["wle"]))


def _wide_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from crt1d_amd.dist import BandShardPlan

        d, bw = _problem_even()
        solve_fn, epi = _oracle_fns()
        plan = BandShardPlan("zq", HostCols(d), HostBands(d), bw, column_tiles=2, solve_fn=solve_fn, epilogue_fn=epi)
        plan()   # ... and again WITHOUT wait(): the pending all-reduces of the first step are completed before their message buffers are
        plan()   # rewritten (re-entry guard), so the second step's result is still the full integral
        r = plan.wait()
        q.put((rank, plan.band_range, {k: r[k].numpy().copy() for k in ("aI", "aI_sl", "aI_sh", "totals", "reflectance")}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [3, 5, 8])
def test_band_partition_even_shards_world_3_5_8(world):
    """An even band count is dealt in PAIRS (band_block_range: every shard even, so every rank runs the 16-byte fused flush), the shards
    tile the spectrum without gap or overlap, and the all-reduced sums reproduce the unsharded ones at 1e-12 on 3, 5 and 8 ranks."""
    from crt1d_amd.dist import band_block_range, solve_sharded

    ranges = [band_block_range(NB_EVEN, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == NB_EVEN
    assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
    assert all((hi - lo) % 2 == 0 and hi > lo for lo, hi in ranges)
    assert max(hi - lo for lo, hi in ranges) - min(hi - lo for lo, hi in ranges) <= 2
    # 300 bands on 8 ranks: 38 x 6 + 36 x 2 (the configuration of BASELINE configs[3])
    r300 = [band_block_range(300, r, 8) for r in range(8)]
    assert sorted(hi - lo for lo, hi in r300) == [36, 36, 38, 38, 38, 38, 38, 38]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wide_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d, bw = _problem_even()
    solve_fn, epi = _oracle_fns()
    ref = solve_sharded("zq", HostCols(d), HostBands(d), bw, partition="column", solve_fn=solve_fn, epilogue_fn=epi)
    for rank, band_range, res in got:
        assert band_range == ranges[rank]
        for k, v in res.items():
            np.testing.assert_allclose(v, ref[k].numpy(), rtol=1e-12, atol=1e-13)
