#!/usr/bin/env python3
"""
bench.py -- the hot path of BASELINE.json measured on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scheme 2s] [--ncol 10000] [--nb 300] [--nz 60] [--partition column|band]

Workload (N = 1): BASELINE.json configs[1] -- ``solve_2s`` batched over 1e4 synthetic profiles x 300 bands x
60 levels, fp64, inputs resident in HBM before the timed region.  One *step* = one pass of the hot path over
the batch = column-precompute kernel K0 + the solve kernel, through the C ABI (``crt_hip_2s_f64``).

N > 1: ``python bench.py --gpus N`` starts N rank processes ITSELF (children of a parent that never touches the GPU, rendezvous
on 127.0.0.1) unless it already runs under ``torch.distributed.run`` (WORLD_SIZE set), one rank per GPU over RCCL.
* ``--partition column`` (default): the (column x band) grid is sharded by column blocks, every rank solves its own ``--ncol``
  columns (weak scaling), no data-path collective (independent units); the timing is max-reduced.  After
  the timed region the ranks all-reduce a vector of ones (``rccl.ranks_seen``) and time the all-reduce of a config-4-sized
  message (``rccl.allreduce_ms``) -- reported, never part of ``value``.
* ``--partition band`` (BASELINE.json configs[3]; defaults zq, 1e5 columns, 100 levels): every rank solves ITS bands of ALL
  columns (K0 replicated), forms partial band sums (``crt_hip_absorb_bandsum_f64``) and the packed sums of every column tile are
  all-reduced over RCCL while the next tile is being solved (``crt1d_amd.dist.BandShardPlan``).  Strong scaling: the total
  work is fixed, ``value`` = ncol x nb x steps / time.

Prints ONE JSON line on rank 0.  ``roofline``: the dominant kernel (the solve kernel, launched alone with
CRT_FLAG_SKIP_PRECOMPUTE between two HIP events on its own stream) against the 8 TB/s HBM3E peak; algorithmic bytes per solve
as SURVEY section 8(d) / BASELINE.md section 3.  The timed block of K steps is repeated ``--repeats`` times: ``ms_per_step`` /
``value`` come from the MEDIAN block, min and max are reported beside it.  ``cpu_baseline``: the NumPy oracle (a port of the
reference algorithm; the reference itself is pure Python and cannot travel to the GPU box) timed on this box's host cores on a
bounded sample of the same workload, rank 0, N = 1 only; ``cpu_baseline.reference_shaped`` is a per-band Python loop with the
reference's structure and the here-measured ratio to the real reference (oracle/ref_ratio.json).  ``pcie_inclusive``: the same
step with H2D of the inputs and D2H of the profiles (never ``value``).
"""

import argparse
import json
import os
import subprocess
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


def bytes_per_solve(scheme, nz, s=8):
    n_in, n_full, n_mid = N_IO[scheme]
    return s * (n_in + n_full * nz + n_mid * (nz - 1))


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): oracle = test infrastructure, used here only as the thing being timed beside the GPU
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
    if sys.stdin.readline().strip() != "go":  # released by run_cpu_workers (EOF = the parent gave up)
        raise SystemExit(3)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s and n < 400:
        fn(oc, **kw)
        n += 1
    return n * chunk * nb, time.perf_counter() - t0


def spawn_cpu_workers(scheme, nb, nz, nproc, budget_s=8.0):
    """Start `nproc` host worker processes of this script (`--cpu-worker`) that import NumPy only, never the GPU, and then BLOCK on their
    stdin until `run_cpu_workers` tells them to go.  Called before this process has imported torch or touched the GPU (process creation
    from a GPU-initialised parent is what the pool's rules warn about), while the timing itself happens after the GPU work is over."""
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker"]
    return [subprocess.Popen(cmd + [scheme, str(nb), str(nz), str(1000 + i), str(budget_s)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, text=True, env=env) for i in range(nproc)]


def stop_cpu_workers(procs):
    for q in procs or []:
        if q.poll() is None:
            q.kill()
    for q in procs or []:
        try:
            q.wait(timeout=10)
        except Exception:
            pass


def run_cpu_workers(procs, budget_s=8.0):
    """Aggregate oracle rate of the waiting worker processes (one per core): release them together, add their rates."""
    total = 0.0
    deadline = time.time() + budget_s + 90
    try:
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        for p in procs:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            solves, secs = out.split()[-2:]
            total += float(solves) / float(secs)
    except Exception:
        stop_cpu_workers(procs)
        raise
    return total


def reference_shaped_baseline(scheme, nb, nz, budget_s=6.0):
    """The reference's structure on this box's host: one column per call and, for 2s, a genuine per-band Python loop
    (oracle/ref_shaped.py).  Returns the rate and what turns it into a reference-equivalent (oracle/ref_ratio.json: the same
    stand-in timed beside the REAL reference in the build container)."""
    from crt1d_amd import synth
    from oracle import crt_oracle as O

    ratios = {}
    try:
        with open(os.path.join(ROOT, "oracle", "ref_ratio.json")) as f:
            ratios = json.load(f)
    except Exception:
        pass
    d = synth.make_columns(20, nb, nz, seed=98)
    oc = O.Columns(d["psi"], d["lai"], mla=d["mla"], g_kind=d["g_kind"], g_param=d["g_param"])
    out = {}
    if scheme == "2s":
        from oracle import ref_shaped

        solves, secs = ref_shaped.time_2s_loop(d, O.mu_bar(oc), oc.K_b(), budget_s=budget_s)
        rate, kind, key = solves / secs, "per-band Python loop, one column per call (oracle/ref_shaped.py: structure of _solve_2s.py:54-156)", "ratio_ref_shaped"
    else:
        kw = dict(I_dr0=d["I_dr0"], I_df0=d["I_df0"], leaf_r=d["leaf_r"], leaf_t=d["leaf_t"], soil_r=d["soil_r"])
        if scheme == "bl":
            kw.pop("soil_r")
        fn = O.SOLVERS[scheme]
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < budget_s:
            c = n % 20
            o1 = O.Columns(d["psi"][c:c + 1], d["lai"][c:c + 1], mla=d["mla"][c:c + 1], g_kind=d["g_kind"][c:c + 1], g_param=d["g_param"][c:c + 1])
            fn(o1, **{k: v[c:c + 1] for k, v in kw.items()})
            n += 1
        rate, kind, key = n * nb / (time.perf_counter() - t0), "oracle called one column at a time (vectorised over bands)", "ratio_oracle_percol"
    out = {"value": rate, "unit": "solves/s", "cores": 1, "what": kind}
    e = ratios.get(scheme, {})
    if key in e and (nb, nz) == (ratios.get("_meta", {}).get("nb"), ratios.get("_meta", {}).get("nz")):
        out["ratio_to_reference"] = e[key]  # stand-in rate / real reference rate, build container, same nb x nz
        out["reference_equivalent"] = rate / e[key]
        out["reference_measured_in_build_container"] = e["reference"]
        out["ratio_source"] = "oracle/ref_ratio.json (oracle/measure_ref_ratio.py: real crt1d.solvers beside the stand-in, 1 core)"
    return out


def cpu_baseline(scheme, nb, nz, budget_s=15.0, workers=None):
    """Oracle (NumPy port of the reference algorithm) on the host: solves/s on 1 core and on the box's cores, bounded sample.
    `workers`: the processes `spawn_cpu_workers` started before the GPU was initialised."""
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
        if el > budget_s * 0.5 or n >= 400:
            break
    one_core = n * chunk * nb / el
    nproc = len(workers) if workers else 1
    all_cores = None
    if nproc > 1:
        try:
            all_cores = run_cpu_workers(workers)
        except Exception as e:  # never lose the bench line over the baseline
            print(f"all-core CPU baseline failed: {e!r}", file=sys.stderr)
    sample = (f"oracle.solve_{scheme} (NumPy port of the reference algorithm, vectorised over {chunk}-column chunks), {nb} bands x {nz} "
              f"levels: 1 core {one_core:.3g} solves/s ({n * chunk} columns in {el:.1f} s); host has {os.cpu_count()} logical cores")
    out = {"value": one_core, "unit": "solves/s", "cores": 1, "kind": "port", "sample": sample}
    if all_cores is not None:
        out.update(value=all_cores, cores=nproc, sample=sample + f"; value = {nproc} independent processes, 8 s each")
    try:
        out["reference_shaped"] = reference_shaped_baseline(scheme, nb, nz, budget_s=min(6.0, budget_s * 0.4))
    except Exception as e:
        print(f"reference-shaped CPU baseline failed: {e!r}", file=sys.stderr)
    return out


def oracle_check(scheme, d, out, ncol, lo=0, hi=None, nsample=8, f32=False, c0=0):
    """Self-check of the TIMED output buffers (outside the timed region): `nsample` columns (first, last, random) of the profiles the
    bench just wrote, against the oracle on the same inputs.  The oracle is test infrastructure and is used here only as the checker.
    `d`: the host inputs of all `ncol` columns; `out`: the device profiles (ncol_out, nz | nz-1, hi - lo); [lo, hi): this rank's bands."""
    import numpy as np

    from oracle import crt_oracle as O

    rng = np.random.default_rng(2718)
    n_out = next(iter(out.values())).shape[0]
    if n_out <= nsample:
        idx = np.arange(n_out)
    else:
        idx = np.unique(np.r_[0, n_out - 1, 1 + rng.choice(n_out - 2, nsample - 2, replace=False)])
    sub = {k: (v[c0 + idx] if hasattr(v, "shape") and v.shape[:1] == (ncol,) else v) for k, v in d.items()}
    if f32:
        sub = {k: (v.astype(np.float32).astype(np.float64) if k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r") else v) for k, v in sub.items()}
    oc = O.Columns(sub["psi"], sub["lai"], mla=sub["mla"], g_kind=sub["g_kind"], g_param=sub["g_param"])
    kw = {k: sub[k][:, lo:hi] for k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r")}
    if scheme == "bl":
        kw.pop("soil_r")
    ref = O.SOLVERS[scheme](oc, **kw)
    worst_p, worst_e, finite = 0.0, 0.0, True
    ti = None
    for k, v in out.items():
        import torch

        if ti is None:
            ti = torch.as_tensor(idx, device=v.device)
        g = v.index_select(0, ti).double().cpu().numpy()
        r = ref[k]
        finite = finite and bool(np.isfinite(g).all())
        scale = np.abs(r).max(axis=1, keepdims=True)
        scale = np.where(scale == 0, 1.0, scale)
        worst_p = max(worst_p, float(np.max(np.abs(g - r) / scale)))
        den = np.maximum(np.abs(r), (1e-6 if f32 else 1e-9) * scale)
        worst_e = max(worst_e, float(np.max(np.abs(g - r) / den)))
    # bars: fp64 profiles 1e-9 of the profile maximum / 1e-6 elementwise (north_star); f32 storage one float rounding
    bar_p, bar_e = (2.0 ** -23, 2.0 ** -23) if f32 else (1e-9, 1e-6)
    return {"against": f"oracle.solve_{scheme} on the same inputs", "columns": [int(c0 + i) for i in idx], "bands": [lo, hi if hi is not None else "all"],
            "arrays": sorted(out), "max_rel_profile": worst_p, "max_rel_elem": worst_e, "finite": finite,
            "bar_profile": bar_p, "bar_elem": bar_e, "ok": bool(finite and worst_p <= bar_p and worst_e <= bar_e)}


def oracle_check_integrated(scheme, d, res, ncol, band_w, nsample=8):
    """The band-integrated results of the timed step (after the all-reduce, all bands) on `nsample` columns against the oracle:
    solve -> layer absorption (model.py:573-647) -> sum over bands with the weights (diagnostics.py:81)."""
    import numpy as np
    import torch

    from oracle import crt_oracle as O

    rng = np.random.default_rng(31415)
    idx = np.arange(ncol) if ncol <= nsample else np.unique(np.r_[0, ncol - 1, 1 + rng.choice(ncol - 2, nsample - 2, replace=False)])
    sub = {k: (v[idx] if hasattr(v, "shape") and v.shape[:1] == (ncol,) else v) for k, v in d.items()}
    oc = O.Columns(sub["psi"], sub["lai"], mla=sub["mla"], g_kind=sub["g_kind"], g_param=sub["g_param"])
    kw = {k: sub[k] for k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r")}
    if scheme == "bl":
        kw.pop("soil_r")
    sol = O.SOLVERS[scheme](oc, **kw)
    ab = O.calc_absorption(oc, sol, leaf_r=sub["leaf_r"], leaf_t=sub["leaf_t"])
    w = np.asarray(band_w)  # (ng, nb)
    ti = torch.as_tensor(idx, device=res["aI"].device)
    worst = 0.0
    for k in ("aI", "aI_sl", "aI_sh"):
        ref = ab[k] @ w.T  # (n, nz-1, ng)
        g = res[k].index_select(0, ti).cpu().numpy()
        worst = max(worst, float(np.max(np.abs(g - ref) / np.abs(ref).max(axis=1, keepdims=True))))
    top_d = (sol["I_dr"][:, -1] + sol["I_df_d"][:, -1]) @ w.T
    top_u = sol["I_df_u"][:, -1] @ w.T
    refl = res["reflectance"].index_select(0, ti).cpu().numpy()
    worst_r = float(np.max(np.abs(refl - top_u / top_d) / np.abs(top_u / top_d)))
    return {"against": f"oracle.solve_{scheme} + calc_absorption + band sums, all bands (after the all-reduce)", "columns": [int(i) for i in idx],
            "max_rel_absorption_profile": worst, "max_rel_reflectance": worst_r, "bar": 1e-9, "ok": bool(worst <= 1e-9 and worst_r <= 1e-9)}


def load_pmc_traffic(kernel, shape):
    """HBM bytes per launch of the solve kernel from the committed rocprofv3 --pmc summaries (separate WRITE_SIZE / FETCH_SIZE passes of
    a bench command with this launch shape, profiles/pmc_traffic.json), if one exists for exactly this kernel family (the name the
    library reports up to its first blank: scheme and storage type included) and launch shape."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    fam = (kernel or "").split(" ")[0]
    try:
        with open(p) as f:
            d = json.load(f)
        for e in d.get("entries", []):
            if (e.get("kernel_reported_by_library") or "").split(" ")[0] == fam and e.get("shape") == list(shape):
                return e.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


# ------------------------------------------------------------------------------------------------------------------
# self-launch: N rank processes as CHILDREN of this (GPU-free) process
def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, timeout_s):
    """Start `n` rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as
    torch.distributed.run would set them) and wait.  Rank 0 inherits stdout (it prints the JSON line); the exit code is the
    first non-zero one.  Nothing here imports torch or touches the GPU, and nothing is exec'ed."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + timeout_s
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                        print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
            if rc != 0 or time.time() > deadline:
                if rc == 0:
                    rc = 124
                    print(f"bench.py: ranks still running after {timeout_s:.0f} s; stopping them", file=sys.stderr)
                break
            time.sleep(0.2)
    finally:
        for p in procs:  # exactly the processes started here
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="steps per timed block (default 100; 10 with --partition band)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps first (default 20; 3 with --partition band)")
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps; the median block is reported")
    ap.add_argument("--scheme", default=None, choices=sorted(N_IO))
    ap.add_argument("--partition", default="column", choices=["column", "band"],
                    help="'column': column blocks per rank, no data-path collective (weak scaling, the headline); 'band': band blocks "
                         "per rank + RCCL all-reduce of the spectral integrals (BASELINE configs[3], strong scaling)")
    ap.add_argument("--column-tiles", type=int, default=4, help="--partition band: column tiles per step (all-reduce of tile i overlaps tile i+1)")
    ap.add_argument("--placement", default="auto", choices=["auto", "none"], help="output arrays from the class-interleaving set allocator "
                    "(crt_hip_buffer_alloc_set) or from torch.empty")
    ap.add_argument("--ncol", type=int, default=None, help="columns PER GPU (column partition; default 10000) or in TOTAL (band partition; default 100000)")
    ap.add_argument("--nb", type=int, default=300)
    ap.add_argument("--nz", type=int, default=None)
    ap.add_argument("--ragged", action="store_true", help="columns with NON-uniform dLAI (SURVEY 8(d): lai = LAI (1 - linspace(0,1,nz)**gamma), gamma ~ U(0.5, 2)); "
                    "the default linspace columns are what every reference LAI generator makes, but any strictly decreasing lai is legal input "
                    "(crt1d/model.py:240-246)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"],
                    help="storage type of spectra/profiles; arithmetic is fp64 either way (f32 = config 5 variant, not the headline)")
    ap.add_argument("--variant", default="profiles", choices=["profiles", "integrated"],
                    help="'integrated': fused solve + absorption + band integrals (crt_hip_integrated_f64), no profiles written; "
                         "a separately reported variant with its own byte count, NOT the headline")
    ap.add_argument("--compare-plain", action="store_true", help="also time the solve kernel on torch.empty outputs (roofline.kernel_ms_avg_torch_empty_outputs); "
                    "off by default so that a rocprofv3 average of the default command describes ONE allocation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle self-check of the timed output buffers")
    ap.add_argument("--no-pcie", action="store_true", help="skip the H2D + step + D2H measurement")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; 'gloo' + --share-device rehearses the N>1 code path on a one-GPU box")
    ap.add_argument("--share-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--launch-check", action="store_true", help="N>1: rendezvous + all-reduce of ones only (no GPU work; CPU test of the launcher)")
    ap.add_argument("--force-pg", action="store_true", help="create the process group and run every collective even with one rank (RCCL API "
                    "rehearsal on a one-GPU box: torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --force-pg)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0)
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    a = ap.parse_args(argv)
    band = a.partition == "band"
    if a.scheme is None:
        a.scheme = "zq" if band else "2s"
    if a.ncol is None:
        a.ncol = 100000 if band else 10000
    if a.nz is None:
        a.nz = 100 if band else 60
    if a.steps is None:
        a.steps = 10 if band else 100
    if a.warmup is None:
        a.warmup = 3 if band else 20
    return a


def kernel_name(lib):
    try:
        return lib.crt_hip_last_kernel().decode()
    except Exception:
        return None


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":  # helper process of cpu_baseline_all_cores
        scheme, nb, nz, seed, budget = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
        solves, secs = _cpu_worker((scheme, nb, nz, seed, budget))
        print(solves, secs)
        return 0
    a = parse_args(sys.argv[1:])
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's command shape is `python bench.py --gpus N ...`: become the launcher (before anything touches the GPU)
        return launch_ranks(a.gpus, sys.argv[1:], a.launch_timeout)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    cpu_workers = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.launch_check:
        nproc = min(16, os.cpu_count() or 1)  # the GPU box's CPU share for one GPU
        if nproc > 1:  # started NOW, before torch is imported or the GPU touched; they idle until the GPU work is over
            cpu_workers = spawn_cpu_workers(a.scheme, a.nb, a.nz, nproc)

    import torch
    import torch.distributed as dist

    if a.launch_check:  # launcher / rendezvous test without a GPU
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": int(t.item()), "master": os.environ.get("MASTER_ADDR")}))
        return 0

    if a.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = world > 1 or a.force_pg  # collectives in use
    if pg:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
    red_dev = dev if a.backend == "nccl" else torch.device("cpu")

    from crt1d_amd import _lib, batched, synth

    lib = _lib.load()
    scheme, ncol, nb, nz = a.scheme, a.ncol, a.nb, a.nz
    stream = torch.cuda.current_stream(dev)
    NG = 3  # PAR, NIR, solar

    def barrier():
        torch.cuda.synchronize(dev)
        if pg:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if pg:
            t = torch.tensor([x], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    def timed_blocks(step, finish=None):
        """--warmup untimed steps, then --repeats blocks of EXACTLY --steps steps, each bracketed by barrier + synchronize on both
        sides; per block the max over ranks."""
        for _ in range(a.warmup):
            step()
        if finish:
            finish()
        blocks = []
        for _ in range(max(1, a.repeats)):
            barrier()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            if finish:
                finish()
            barrier()
            blocks.append(max_over_ranks(time.perf_counter() - t0))
        return blocks

    f32 = a.dtype == "f32"
    band = a.partition == "band"
    extra = {}
    check, check_int = None, None
    if band:
        # ---------------------------------------------------------------- band partition: BASELINE configs[3]
        from crt1d_amd import dist as cdist
        from crt1d_amd import spectra

        if f32:
            raise SystemExit("--partition band runs the fp64 path")
        d = synth.make_columns(ncol, nb, nz, seed=1234, uniform_dlai=not a.ragged)  # the SAME columns on every rank: K0 is replicated, the bands are sharded
        cols = batched.Columns.from_host(d, dev)
        bands = batched.Bands.from_host(d, dev)
        bw = torch.as_tensor(spectra.band_weights(d["wle"])).to(dev)
        # --variant integrated: the fused kernel (crt_hip_integrated_f64) forms the band sums without writing any profile
        plan = cdist.BandShardPlan(scheme, cols, bands, bw, column_tiles=a.column_tiles, share_profiles=True, placement=a.placement,
                                   keep_profiles=a.variant == "profiles", always_reduce=a.force_pg)
        del bands
        nb_local = plan.band_range[1] - plan.band_range[0]
        res_holder = {}

        def step():
            plan()
            res_holder["r"] = plan.wait()

        blocks = timed_blocks(step)
        solves_per_step = ncol * nb  # the whole problem, whatever N is
        # compute only (no collective), and the collective alone on the same messages
        def compute_only():
            plan(reduce=False)

        compute_blocks = timed_blocks(compute_only)
        coll_ms, ranks_seen = None, 1
        if pg:
            ones = torch.ones(1, dtype=torch.float64, device=red_dev)
            dist.all_reduce(ones)
            ranks_seen = int(ones.item())
            barrier()
            t0 = time.perf_counter()
            for _ in range(5):
                for tile in plan.tiles:
                    dist.all_reduce(tile.flat)
            barrier()
            coll_ms = max_over_ranks((time.perf_counter() - t0) / 5 * 1e3)
        kname = kernel_name(lib)
        knames = [kname]
        if pg:
            knames = [None] * world
            dist.all_gather_object(knames, (rank, plan.band_range, kname))
        ref = res_holder["r"]["reflectance"]
        extra = {
            "partition": "band blocks per rank (K0 replicated) + packed fp64 all-reduce per column tile",
            "bands_this_rank": nb_local,
            "column_tiles": plan.ntile,
            "collective": {
                "backend": "RCCL (nccl)" if a.backend == "nccl" else "gloo (rehearsal)", "ranks_seen": ranks_seen,
                "message_bytes_per_step": plan.message_bytes, "messages_per_step": plan.ntile,
                "collective_ms_alone": coll_ms,
                "compute_only_ms_per_step": sorted(compute_blocks)[len(compute_blocks) // 2] / a.steps * 1e3,
                "semantics": "sum over bands of w * X (crt1d/diagnostics.py:81); reflectance = reflected / incoming after the reduce (:510-511)",
            },
            "kernels_per_rank": knames,
            "check_sums": {"mean_solar_reflectance": float(ref[:, 2].mean()), "finite": bool(torch.isfinite(ref).all())},
        }
        if not a.no_check:
            # self-check of the timed step's buffers: (i) the profiles the LAST column tile left in the (shared) output set, this rank's bands;
            # (ii) the integrated results of ALL columns after the all-reduce (rank 0: they are complete on every rank)
            try:
                lt = plan.tiles[-1]
                if a.variant == "profiles" and lt.profiles is not None:
                    check = oracle_check(scheme, d, lt.profiles, ncol, plan.band_range[0], plan.band_range[1], c0=lt.clo)
                else:
                    check = None
                check_int = oracle_check_integrated(scheme, d, res_holder["r"], ncol, spectra.band_weights(d["wle"])) if rank == 0 else None
            except Exception as e:  # the check must never cost the bench line; a failure is reported in it
                check, check_int = {"ok": False, "error": repr(e)}, None
        main_plan = plan.tiles[0].kernel_plan
        ncol_kernel = plan.tiles[0].chi - plan.tiles[0].clo
        nb_kernel = nb_local
    else:
        # ---------------------------------------------------------------- column partition: the headline
        d = synth.make_columns(ncol, nb, nz, seed=1234 + rank, uniform_dlai=not a.ragged)  # this rank's column block: its own seed -> distinct columns of one grid
        cols = batched.Columns.from_host(d, dev)
        d64_host = d
        if f32:
            import numpy as np

            d = {k: (v.astype(np.float32) if k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r") else v) for k, v in d.items()}
        bands = batched.Bands.from_host(d, dev)
        if a.variant == "integrated":
            from crt1d_amd import spectra

            main_plan = batched.IntegratedPlan(scheme, cols, bands, torch.as_tensor(spectra.band_weights(d["wle"])).to(dev))
        else:
            # output buffers are allocated once, before the timed region (placement="auto": crt_hip_buffer_alloc_set)
            main_plan = batched.Plan(scheme, cols, bands, placement=a.placement)
        blocks = timed_blocks(lambda: main_plan())
        solves_per_step = ncol * nb * world
        if not a.no_check:
            try:  # the buffers as the timed region left them, before any probe overwrites them
                if a.variant == "profiles":
                    check = oracle_check(scheme, d64_host if f32 else d, main_plan.out, ncol, f32=f32)
                elif rank == 0:
                    r_int = dict(main_plan.out)
                    r_int["reflectance"] = r_int["totals"][..., 1] / r_int["totals"][..., 0]
                    check_int = oracle_check_integrated(scheme, d, r_int, ncol, spectra.band_weights(d["wle"]))
            except Exception as e:
                check = {"ok": False, "error": repr(e)}
        ncol_kernel, nb_kernel = ncol, nb
        if pg:
            # RCCL evidence for the scaling runs (outside the timed region, never part of `value`): did the collective see N ranks,
            # and what does the all-reduce of one config-4 column tile's packed sums (2.5e4 x (99 x 6 + 12) doubles = 121 MB) cost here
            ones = torch.ones(1, dtype=torch.float64, device=red_dev)
            dist.all_reduce(ones)
            msg = torch.zeros(25000 * (99 * 6 + 12), dtype=torch.float64, device=red_dev)
            dist.all_reduce(msg)
            barrier()
            t0 = time.perf_counter()
            for _ in range(5):
                dist.all_reduce(msg)
            barrier()
            ar_ms = max_over_ranks((time.perf_counter() - t0) / 5 * 1e3)
            knames = [None] * world
            dist.all_gather_object(knames, (rank, kernel_name(lib)))
            # ... and the column partition's own final reduce: grid means of the spectrally integrated absorption / reflectance over the
            # columns of ALL ranks (crt1d_amd.dist.grid_mean: local sums, one small all-reduce, ratios after it)
            gm = None
            if a.variant == "profiles" and not f32:
                try:
                    from crt1d_amd import dist as cdist
                    from crt1d_amd import spectra

                    bwt = torch.as_tensor(spectra.band_weights(d["wle"])).to(dev)
                    loc = batched.absorb_bandsum(cols, bands, main_plan.out, bwt)
                    if a.backend != "nccl":
                        loc = {k: v.cpu() for k, v in loc.items()}
                    g = cdist.grid_mean(loc, ncol * world)
                    gm = {"columns": ncol * world, "reflectance_PAR_NIR_solar": [float(x) for x in g["reflectance"]],
                          "canopy_absorbed_solar_Wm2": float(g["aI"][:, 2].sum()), "message_bytes": int(8 * sum(v[0].numel() for v in loc.values()))}
                except Exception as e:
                    print(f"grid-mean reduce failed: {e!r}", file=sys.stderr)
            extra["rccl"] = {"backend": "RCCL (nccl)" if a.backend == "nccl" else "gloo (rehearsal)", "ranks_seen": int(ones.item()),
                             "allreduce_ms": ar_ms, "allreduce_bytes": msg.numel() * 8,
                             "busbw_GBs": 2 * (world - 1) / world * msg.numel() * 8 / (ar_ms * 1e-3) / 1e9 if world > 1 else None, "kernels_per_rank": knames,
                             "grid_mean": gm,
                             "note": "measured after the timed region; the column partition has no data-path collective in the solve; grid_mean is its final "
                                     "reduce of the integrated quantities"}
            del msg

    blocks_sorted = sorted(blocks)
    el = blocks_sorted[len(blocks_sorted) // 2]  # median block
    value = solves_per_step * a.steps / el

    # ---- dominant kernel alone, HIP events on the stream it is launched on ----
    roof = None
    if main_plan is not None:
        reps = max(20, min(a.steps, 100))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for _ in range(3):
            main_plan(flags=_lib.FLAG_SKIP_PRECOMPUTE)
        torch.cuda.synchronize(dev)
        for e0, e1 in evs:
            e0.record(stream)
            main_plan(stream, flags=_lib.FLAG_SKIP_PRECOMPUTE)
            e1.record(stream)
        torch.cuda.synchronize(dev)
        kname = kernel_name(lib)
        kt = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        k_ms_avg = sum(kt) / len(kt)
        k_ms_med = kt[len(kt) // 2]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            main_plan(stream, flags=_lib.FLAG_PRECOMPUTE_ONLY)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        k0_ms = e0.elapsed_time(e1) / 10

        # the same kernel on plain torch.empty outputs (whatever the driver hands out: the round-1 "placement lottery"), for comparison
        k_ms_plain = None
        if a.compare_plain and world == 1 and not band and a.variant == "profiles" and a.placement == "auto" and getattr(main_plan, "placement_report", None):
            try:
                plain = batched.Plan(scheme, cols, bands, placement="none", workspace=main_plan.workspace)
                for _ in range(3):
                    plain(flags=_lib.FLAG_SKIP_PRECOMPUTE)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(20):
                    plain(stream, flags=_lib.FLAG_SKIP_PRECOMPUTE)
                e1.record(stream)
                torch.cuda.synchronize(dev)
                k_ms_plain = e0.elapsed_time(e1) / 20
                del plain
            except Exception as e:
                print(f"plain-allocation comparison failed: {e!r}", file=sys.stderr)

        bps = bytes_per_solve(scheme, nz, 4 if f32 else 8)
        if a.variant == "integrated":  # B = s n_in + s n_red (nz-1) / nb  (SURVEY 8(d)); n_red = 3 quantities x NG groups
            bps = 8 * N_IO[scheme][0] + 8 * 3 * NG * (nz - 1) / nb
        alg_bytes = bps * ncol_kernel * nb_kernel  # per launch, this GPU
        achieved = alg_bytes / (k_ms_avg * 1e-3) / 1e9

        # ---- measured HBM ceilings from the same run (streaming fill / copy of 4 GiB) ----
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
        # the flush of the solve kernel alone (streaming stores, all arrays in step) on the plan's OWN output arrays: the store rate this
        # placement allows.  fp64 profile outputs only; the arrays are overwritten, so the plan is run once more afterwards.
        store_set_gbs = None
        outs = getattr(main_plan, "out", None)
        if a.variant == "profiles" and not f32 and isinstance(outs, dict):
            full = [v for v in outs.values() if v.dtype == torch.float64 and tuple(v.shape) == (ncol_kernel, nz, nb_kernel)]
            if full and (nz * nb_kernel) % 2 == 0:
                import ctypes

                ptrs = (ctypes.c_void_p * len(full))(*[v.data_ptr() for v in full])
                import math

                t_run = 16 // math.gcd(nb_kernel, 16)  # levels per run: the shortest whole number of 128-B lines, at least 4 levels
                while t_run < 4:
                    t_run *= 2
                run = t_run * nb_kernel
                t_set = timed(lambda: lib.crt_hip_probe_store_set_f64(ptrs, len(full), ncol_kernel, nz * nb_kernel, run, 0.5, stream.cuda_stream))
                store_set_gbs = 8 * len(full) * ncol_kernel * nz * nb_kernel / t_set / 1e9
                main_plan()
                torch.cuda.synchronize(dev)
        roof = {
            "bound": "hbm",
            "kernel": kname,
            "note": None if a.variant == "profiles" else "integrated variant is compute-bound: the HBM fraction is informational only",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": load_pmc_traffic(kname, (ncol_kernel, nb_kernel, nz)),
            "traffic_source": "lookup in profiles/pmc_traffic.json: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of this command (same kernel family "
                              "and launch shape, tools/profile_round.sh), NOT a counter of this run; null when that shape was not profiled",
            "algorithmic_bytes_per_launch": alg_bytes,
            "bytes_per_solve": bps,
            "launch_shape": [ncol_kernel, nb_kernel, nz],
            "kernel_ms_avg": k_ms_avg,
            "kernel_ms_median": k_ms_med,
            "kernel_ms_min": kt[0],
            "kernel_ms_max": kt[-1],
            "kernel_ms_avg_torch_empty_outputs": k_ms_plain,
            "k0_ms": k0_ms,
            # the meaningful yardstick: the kernel's own flush pattern (no arithmetic) on the kernel's own output arrays
            "measured_store_set_GBs": store_set_gbs,
            "frac_of_measured_store_set": (achieved / store_set_gbs) if store_set_gbs else None,
            # reference points only, NOT ceilings: a linear fill writes one memory class at a time and is slower than the class-interleaved
            # column pattern; a copy adds the read stream
            "probe_linear_fill_GBs": fill_gbs,
            "probe_copy_GBs": copy_gbs,
        }

    # ---- PCIe-inclusive step (N = 1, column partition): H2D of the per-band inputs + K0 + solve + D2H of every profile ----
    pcie = None
    if world == 1 and not band and a.variant == "profiles" and not a.no_pcie and rank == 0:
        try:
            out_bytes = sum(v.numel() * v.element_size() for v in main_plan.out.values())
            if out_bytes <= (24 << 30):
                host_in = {k: getattr(bands, k).cpu().pin_memory() for k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r") if getattr(bands, k) is not None}
                host_out = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in main_plan.out.items()}
                in_bytes = sum(v.numel() * v.element_size() for v in host_in.values())

                def pcie_step():
                    for k, v in host_in.items():
                        getattr(bands, k).copy_(v, non_blocking=True)
                    main_plan()
                    for k, v in main_plan.out.items():
                        host_out[k].copy_(v, non_blocking=True)

                pcie_step()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(3):
                    pcie_step()
                torch.cuda.synchronize(dev)
                pt = (time.perf_counter() - t0) / 3
                pcie = {"value": ncol * nb / pt, "unit": "solves/s", "ms_per_step": pt * 1e3, "h2d_bytes": in_bytes, "d2h_bytes": out_bytes,
                        "GBs": (in_bytes + out_bytes) / pt / 1e9, "note": "pinned host buffers; transfer-bound; never `value`"}
                del host_in, host_out
        except Exception as e:
            print(f"pcie-inclusive measurement failed: {e!r}", file=sys.stderr)

    per_step = [b / a.steps * 1e3 for b in blocks]
    out = {
        "metric": ("(column x band) solves/sec at 60 layers" if nz == 60 else f"(column x band) solves/sec at {nz} layers")
                  + (" [integrated outputs only]" if a.variant == "integrated" else ""),
        "value": value,
        "unit": "solves/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": el / a.steps * 1e3,
        "ms_per_step_blocks": {"repeats": len(blocks), "median": el / a.steps * 1e3, "min": min(per_step), "max": max(per_step), "all": per_step},
        "higher_is_better": True,
        "scaling": "strong" if band else "weak",
        "vs_baseline": None,
        "dtype": "f64" if not f32 else "f64 arithmetic, f32 storage",
        "data": "synthetic",
        "config": {
            "workload": (f"solve_{scheme} batched: {ncol} synthetic profiles x {nb} bands x {nz} levels " + ("in total, band-sharded" if band else "per GPU")
                         + ", fp64" + (" (BASELINE.json configs[1] shape)" if (scheme, ncol, nb, nz, band) == ("2s", 10000, 300, 60, False) else "")
                         + (" (BASELINE.json configs[3] shape)" if (scheme, ncol, nb, nz, band) == ("zq", 100000, 300, 100, True) else "")),
            "scheme": scheme, ("ncol_total" if band else "ncol_per_gpu"): ncol, "nb": nb, "nz": nz,
            "dlai": "ragged (lai = LAI (1 - x**gamma), gamma ~ U(0.5, 2))" if a.ragged else "uniform (linspace, as every reference LAI generator)",
            "partition": extra.pop("partition", "column blocks, no collective"),
            "step": (("K0 column precompute + solve kernel via crt_hip_%s_f64" % scheme) + (" + crt_hip_absorb_bandsum_f64 + all-reduce, per column tile" if band else ""))
                    if a.variant == "profiles" else ("K0 + crt_hip_integrated_f64 (no profiles written)" + (" + all-reduce, per column tile" if band else "")),
            "output_placement": getattr(main_plan, "placement_report", None) if a.placement == "auto" else "torch.empty (--placement none)",
        },
    }
    out["config"].update(extra)
    if roof is not None:
        out["roofline"] = roof
    if pcie is not None:
        out["pcie_inclusive"] = pcie
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scheme, nb, nz, a.cpu_budget, cpu_workers)
    stop_cpu_workers(cpu_workers)
    if not a.no_check:  # (rank-independent condition: every rank takes part in the gather)
        checks = [check]
        if pg:  # every rank checked its own buffers: rank 0 reports all of them
            checks = [None] * world
            dist.all_gather_object(checks, check)
        out["check"] = checks[0] if len(checks) == 1 else {"ok": all(c is None or c.get("ok") for c in checks), "per_rank": checks}
        if checks == [None]:
            out.pop("check")
        if check_int is not None:
            out["check_integrated"] = check_int
    if pg:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
