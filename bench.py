#!/usr/bin/env python3
"""
bench.py -- the hot path of BASELINE.json measured on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scheme 2s] [--ncol 10000] [--nb 300] [--nz 60]

Workload (N = 1): BASELINE.json configs[1] -- ``solve_2s`` batched over 1e4 synthetic profiles x 300 bands x
60 levels, fp64, inputs resident in HBM before the timed region.  One *step* = one pass of the hot path over
the batch = column-precompute kernel K0 + the solve kernel, through the C ABI (``crt_hip_2s_f64``).
N > 1 (``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...``): the (column x band) grid is
sharded by column blocks, every rank solves its own ``--ncol`` columns (weak scaling), no data-path collective
(independent units); only the timing is reduced (max over ranks).

Prints ONE JSON line on rank 0.  ``roofline``: the dominant kernel (the solve kernel, launched alone with
CRT_FLAG_SKIP_PRECOMPUTE between two HIP events on its own stream) against the 8 TB/s HBM3E peak;
algorithmic bytes per solve as SURVEY section 8(d) / BASELINE.md section 3.  ``cpu_baseline``: the NumPy oracle
(a port of the reference algorithm; the reference itself is pure Python and cannot travel to the GPU box)
timed on this box's host cores on a bounded sample of the same workload, rank 0, N = 1 only.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md

# algorithmic bytes per (column x band) solve: s * [n_in + n_full * nz + n_mid * (nz - 1)], s = 8 (fp64)
# (each compulsory byte once: per-band inputs read, full profiles written)
N_IO = {  # scheme: (n_in, n_full, n_mid)
    "2s": (5, 4, 0), "4s": (5, 4, 0), "bl": (4, 4, 0), "g77": (5, 7, 0), "bf": (5, 7, 0), "n79": (5, 4, 2), "zq": (5, 7, 0),
    "zq_pa": (5, 4, 0),
}
# dominant kernel per scheme as rocprofv3 names it (template arguments abbreviated)
KERNEL_NAMES = {"2s": "k_pipe<Sch2s>", "4s": "k_pipe<Sch4s>", "bl": "k_pipe<SchBl>", "g77": "k_tile<SchG77<false>>",
                "bf": "k_tile<SchG77<true>>", "n79": "k_tri_pipe<TriN79>", "zq": "k_tri_pipe<TriZq>", "zq_pa": "k_zqpa_pipe (grid solve + interpolation in one kernel)"}


def bytes_per_solve(scheme, nz, s=8):
    n_in, n_full, n_mid = N_IO[scheme]
    return s * (n_in + n_full * nz + n_mid * (nz - 1))


def _cpu_worker(args):
    """One host process: time the oracle on its own chunk (spawned, never touches the GPU)."""
    scheme, nb, nz, seed, budget_s = args
    import numpy as np  # noqa: F401

    from crt1d_amd import synth
    from oracle import crt_oracle as O

    chunk = 50
    d = synth.make_columns(chunk, nb, nz, seed=seed)
    oc = O.Columns(d["psi"], d["lai"], mla=d["mla"], g_kind=d["g_kind"], g_param=d["g_param"])
    kw = dict(I_dr0=d["I_dr0"], I_df0=d["I_df0"], leaf_r=d["leaf_r"], leaf_t=d["leaf_t"], soil_r=d["soil_r"])
    if scheme == "bl":
        kw.pop("soil_r")
    fn = O.SOLVERS[scheme]
    fn(oc, **kw)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s and n < 400:
        fn(oc, **kw)
        n += 1
    return n * chunk * nb, time.perf_counter() - t0


def cpu_baseline_all_cores(scheme, nb, nz, nproc, budget_s=8.0):
    """Aggregate oracle rate of `nproc` independent host processes (one per core).
    Plain subprocesses of this script (`--cpu-worker`), each with a hard timeout: they import NumPy only, never the GPU."""
    import subprocess

    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker"]
    procs = [subprocess.Popen(cmd + [scheme, str(nb), str(nz), str(1000 + i), str(budget_s)], stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True, env=env) for i in range(nproc)]
    total = 0.0
    deadline = time.time() + budget_s + 90
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            solves, secs = out.split()[-2:]
            total += float(solves) / float(secs)
        except Exception:
            p.kill()
            raise
    return total


def cpu_baseline(scheme, nb, nz, budget_s=15.0):
    """Oracle (NumPy port of the reference algorithm) on the host: solves/s on 1 core, bounded sample."""
    import numpy as np

    from crt1d_amd import synth
    from oracle import crt_oracle as O

    chunk = 50
    d = synth.make_columns(chunk, nb, nz, seed=99)
    oc = O.Columns(d["psi"], d["lai"], mla=d["mla"], g_kind=d["g_kind"], g_param=d["g_param"])
    kw = dict(I_dr0=d["I_dr0"], I_df0=d["I_df0"], leaf_r=d["leaf_r"], leaf_t=d["leaf_t"], soil_r=d["soil_r"])
    if scheme == "bl":
        kw.pop("soil_r")
    fn = O.SOLVERS[scheme]
    fn(oc, **kw)  # warm
    t0 = time.perf_counter()
    n = 0
    while True:
        fn(oc, **kw)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 400:
            break
    # reference-shaped variant: one column per call (the reference has no batching), a few columns only
    t1 = time.perf_counter()
    ncs = 0
    for c in range(min(chunk, 20)):
        o1 = O.Columns(d["psi"][c:c + 1], d["lai"][c:c + 1], mla=d["mla"][c:c + 1], g_kind=d["g_kind"][c:c + 1],
                       g_param=d["g_param"][c:c + 1])
        fn(o1, **{k: v[c:c + 1] for k, v in kw.items()})
        ncs += 1
    per_col = (time.perf_counter() - t1) / ncs
    one_core = n * chunk * nb / el
    nproc = min(16, os.cpu_count() or 1)  # the GPU box's CPU share for one GPU
    all_cores = None
    if nproc > 1:
        try:
            all_cores = cpu_baseline_all_cores(scheme, nb, nz, nproc)
        except Exception as e:  # never lose the bench line over the baseline
            print(f"all-core CPU baseline failed: {e!r}", file=sys.stderr)
    sample = (f"oracle.solve_{scheme} (NumPy port of the reference algorithm, vectorised over {chunk}-column chunks), {nb} bands x {nz} "
              f"levels: 1 core {one_core:.3g} solves/s ({n * chunk} columns in {el:.1f} s); one-column-per-call (reference-shaped) "
              f"{nb / per_col:.3g} solves/s on 1 core; host has {os.cpu_count()} logical cores")
    if all_cores is not None:
        return {"value": all_cores, "unit": "solves/s", "cores": nproc, "kind": "port",
                "sample": sample + f"; value = {nproc} independent processes, 8 s each"}
    return {"value": one_core, "unit": "solves/s", "cores": 1, "kind": "port", "sample": sample}


def load_pmc_traffic(scheme, ncol, nb, nz):
    """HBM bytes per launch of the solve kernel from the committed rocprofv3 --pmc summary (separate WRITE_SIZE /
    FETCH_SIZE passes of this same command, profiles/), if one exists for exactly this configuration."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            d = json.load(f)
        e = d.get(scheme)
        if e is None or e.get("shape") != [ncol, nb, nz]:
            return None
        return e.get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":  # helper process of cpu_baseline_all_cores
        scheme, nb, nz, seed, budget = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
        solves, secs = _cpu_worker((scheme, nb, nz, seed, budget))
        print(solves, secs)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scheme", default="2s", choices=sorted(N_IO))
    ap.add_argument("--placement", default="auto", choices=["auto", "none"], help="output-buffer placement search at plan creation")
    ap.add_argument("--ncol", type=int, default=10000, help="columns PER GPU")
    ap.add_argument("--nb", type=int, default=300)
    ap.add_argument("--nz", type=int, default=60)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"],
                    help="storage type of spectra/profiles; arithmetic is fp64 either way (f32 = config 5 variant, not the headline)")
    ap.add_argument("--variant", default="profiles", choices=["profiles", "integrated"],
                    help="'integrated': fused solve + absorption + band integrals (crt_hip_integrated_f64), no profiles written; "
                         "a separately reported variant with its own byte count, NOT the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; 'gloo' + --share-device rehearses the N>1 code path on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit(f"--gpus {a.gpus} needs: python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py ...")
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
    red_dev = dev if a.backend == "nccl" else torch.device("cpu")

    from crt1d_amd import _lib, batched, synth

    scheme, ncol, nb, nz = a.scheme, a.ncol, a.nb, a.nz
    # column shard of this rank: its own seed -> distinct columns of one global grid
    d = synth.make_columns(ncol, nb, nz, seed=1234 + rank)
    cols = batched.Columns.from_host(d, dev)
    if a.dtype == "f32":
        import numpy as np

        d = {k: (v.astype(np.float32) if k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r") else v) for k, v in d.items()}
    bands = batched.Bands.from_host(d, dev)
    NG = 3  # PAR, NIR, solar
    if a.variant == "integrated":
        from crt1d_amd import spectra

        plan = batched.IntegratedPlan(scheme, cols, bands, torch.as_tensor(spectra.band_weights(d["wle"])).to(dev))
    else:
        # output buffers are allocated once, before the timed region; placement="auto" lets the plan pick where they live
        # (the same kernel runs ~15 % faster or slower depending on where the driver put them)
        plan = batched.Plan(scheme, cols, bands, placement=a.placement)
    stream = torch.cuda.current_stream(dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        plan()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        plan()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    solves_per_step = ncol * nb * world
    value = solves_per_step * a.steps / el

    # ---- dominant kernel alone, HIP events on the stream it is launched on ----
    reps = max(20, a.steps)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for _ in range(3):
        plan(flags=_lib.FLAG_SKIP_PRECOMPUTE)
    torch.cuda.synchronize(dev)
    for e0, e1 in evs:
        e0.record(stream)
        plan(stream, flags=_lib.FLAG_SKIP_PRECOMPUTE)
        e1.record(stream)
    torch.cuda.synchronize(dev)
    kt = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    k_ms_avg = sum(kt) / len(kt)
    k_ms_med = kt[len(kt) // 2]
    # K0 alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10):
        plan(stream, flags=_lib.FLAG_PRECOMPUTE_ONLY)
    e1.record(stream)
    torch.cuda.synchronize(dev)
    k0_ms = e0.elapsed_time(e1) / 10

    bps = bytes_per_solve(scheme, nz, 4 if a.dtype == "f32" else 8)
    if a.variant == "integrated":  # B = s n_in + s n_red (nz-1) / nb  (SURVEY 8(d)); n_red = 3 quantities x NG groups
        bps = 8 * N_IO[scheme][0] + 8 * 3 * NG * (nz - 1) / nb
    alg_bytes = bps * ncol * nb  # per launch, this GPU
    achieved = alg_bytes / (k_ms_avg * 1e-3) / 1e9

    # ---- measured HBM ceilings from the same run (streaming fill / copy of 4 GiB) ----
    lib = _lib.load()
    n = 1 << 29  # doubles = 4 GiB
    buf = torch.empty(n, dtype=torch.float64, device=dev)
    src = torch.empty(n, dtype=torch.float64, device=dev).fill_(1.0)

    def timed(fn, nrep=5):
        fn()
        torch.cuda.synchronize(dev)
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(stream)
        for _ in range(nrep):
            fn()
        a1.record(stream)
        torch.cuda.synchronize(dev)
        return a0.elapsed_time(a1) / nrep * 1e-3

    t_fill = timed(lambda: lib.crt_hip_probe_fill_f64(buf.data_ptr(), n, 0.5, stream.cuda_stream))
    t_copy = timed(lambda: lib.crt_hip_probe_copy_f64(buf.data_ptr(), src.data_ptr(), n, stream.cuda_stream))
    fill_gbs = 8 * n / t_fill / 1e9
    copy_gbs = 2 * 8 * n / t_copy / 1e9
    del buf, src

    out = {
        "metric": ("(column x band) solves/sec at 60 layers" if nz == 60 else f"(column x band) solves/sec at {nz} layers")
                  + (" [integrated outputs only]" if a.variant == "integrated" else ""),
        "value": value,
        "unit": "solves/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": el / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64" if a.dtype == "f64" else "f64 arithmetic, f32 storage",
        "data": "synthetic",
        "config": {
            "workload": f"solve_{scheme} batched: {ncol} synthetic profiles x {nb} bands x {nz} levels per GPU, fp64 "
                        f"(BASELINE.json configs[1] shape)" if (scheme, ncol, nb, nz) == ("2s", 10000, 300, 60) else
                        f"solve_{scheme} batched: {ncol} synthetic profiles x {nb} bands x {nz} levels per GPU, fp64",
            "scheme": scheme, "ncol_per_gpu": ncol, "nb": nb, "nz": nz, "partition": "column blocks, no collective",
            "step": "K0 column precompute + solve kernel via crt_hip_%s_f64" % scheme,
            "output_placement": getattr(plan, "placement_report", None),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": KERNEL_NAMES[scheme] if a.variant == "profiles" else ("k_int" if scheme in ("2s", "4s", "bl", "g77", "bf") else "k_tri_int"),
            "note": None if a.variant == "profiles" else "integrated variant is compute-bound: the HBM fraction is informational only",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": load_pmc_traffic(scheme, ncol, nb, nz) if (a.dtype == "f64" and a.variant == "profiles") else None,
            "algorithmic_bytes_per_launch": alg_bytes,
            "bytes_per_solve": bps,
            "kernel_ms_avg": k_ms_avg,
            "kernel_ms_median": k_ms_med,
            "k0_ms": k0_ms,
            "measured_fill_GBs": fill_gbs,
            "measured_copy_GBs": copy_gbs,
            "frac_of_measured_fill": achieved / fill_gbs,
        },
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scheme, nb, nz, a.cpu_budget)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
