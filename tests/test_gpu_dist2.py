"""GPU, two rank PROCESSES sharing the one GPU of the box (gloo process group; RCCL refuses two ranks on one device): the planned
band partition on the real HIP kernels -- crt1d_amd.dist.BandShardPlan with its Plan / BandSumPlan per column tile, packed messages
and asynchronous all-reduces -- must give every rank the integrated results of the unsharded problem."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NCOL, NB, NZ = 50, 300, 40


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, scheme, keep_profiles, q):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from crt1d_amd import batched, spectra, synth
        from crt1d_amd.dist import BandShardPlan

        d = synth.make_columns(NCOL, NB, NZ, seed=3)
        cols, bands = batched.Columns.from_host(d, "cuda:0"), batched.Bands.from_host(d, "cuda:0")
        bw = torch.as_tensor(spectra.band_weights(d["wle"])).cuda()
        plan = BandShardPlan(scheme, cols, bands, bw, column_tiles=3, share_profiles=True, keep_profiles=keep_profiles)
        plan().wait()
        r = plan().wait()  # second step on the same buffers
        torch.cuda.synchronize()
        q.put((rank, plan.band_range, {k: r[k].cpu().numpy() for k in ("aI", "aI_sl", "aI_sh", "totals", "reflectance")}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scheme,keep_profiles", [("zq", True), ("2s", True), ("n79", False)])
def test_band_shard_plan_two_ranks_on_the_hip_kernels(scheme, keep_profiles):
    import torch
    import torch.multiprocessing as mp

    from crt1d_amd import batched, spectra, synth
    from crt1d_amd.dist import solve_sharded

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, scheme, keep_profiles, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = synth.make_columns(NCOL, NB, NZ, seed=3)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    bw = torch.as_tensor(spectra.band_weights(d["wle"])).cuda()
    ref = solve_sharded(scheme, cols, bands, bw, partition="column")  # world of one: the whole problem, HIP kernels
    assert [g[1] for g in got] == [(0, 150), (150, 300)]
    scale = float(ref["totals"].abs().max())
    for rank, _, res in got:
        for k in ("aI", "aI_sl", "aI_sh", "totals"):
            # two partial band sums added by the all-reduce instead of one sum over 300 bands: rounding only
            assert np.max(np.abs(res[k] - ref[k].cpu().numpy())) <= 1e-12 * scale, (rank, k)
        np.testing.assert_allclose(res["reflectance"], ref["reflectance"].cpu().numpy(), rtol=1e-12)
    for k in got[0][2]:
        np.testing.assert_array_equal(got[0][2][k], got[1][2][k])  # both ranks hold the same reduced result
