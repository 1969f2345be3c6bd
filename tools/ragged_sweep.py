"""Uniform vs ragged dLAI, every scheme, kernel alone (HIP events on the launch stream) and K0 alone.

    python tools/ragged_sweep.py [ncol nb nz] [--schemes 2s,4s,...] [--tune k=v,...]

Prints one line per (scheme, column kind): kernel ms (median of 5 rounds x 10 launches), fraction of 8 TB/s at the
algorithmic byte count (bench.bytes_per_solve), K0 ms, the kernel the library chose."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from crt1d_amd import _lib, batched, synth


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    ncol, nb, nz = (int(x) for x in args[:3]) if len(args) >= 3 else (10000, 300, 60)
    schemes = opts.get("schemes", "2s,4s,bl,g77,bf,n79,zq,zq_pa").split(",")
    tune = {int(k): int(v) for k, v in (kv.split(":") for kv in opts["tune"].split(","))} if "tune" in opts else {}
    f32 = opts.get("dtype") == "f32"
    st = torch.cuda.current_stream()
    for scheme in schemes:
        plans = {}
        out = None
        for kind in ("uniform", "ragged"):
            d = synth.make_columns(ncol, nb, nz, seed=1234, uniform_dlai=kind == "uniform")
            cols = batched.Columns.from_host(d)
            if f32:
                import numpy as np

                d = {k: (v.astype(np.float32) if k in ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r") else v) for k, v in d.items()}
            bands = batched.Bands.from_host(d)
            p = batched.Plan(scheme, cols, bands, out=out, tune=tune)  # both kinds write the SAME output set (same placement)
            out = p.out
            p()
            torch.cuda.synchronize()
            plans[kind] = p
        res = {k: [] for k in plans}
        k0 = {}
        for _ in range(5):
            for kind, p in plans.items():
                p(flags=_lib.FLAG_SKIP_PRECOMPUTE)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(10):
                    p(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
                e1.record(st)
                torch.cuda.synchronize()
                res[kind].append(e0.elapsed_time(e1) / 10)
        for kind, p in plans.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                p(st, flags=_lib.FLAG_PRECOMPUTE_ONLY)
            e1.record(st)
            torch.cuda.synchronize()
            k0[kind] = e0.elapsed_time(e1) / 10
        bps = bench.bytes_per_solve(scheme, nz, 4 if f32 else 8)
        for kind, v in res.items():
            v = sorted(v)
            ms = v[len(v) // 2]
            plans[kind](flags=_lib.FLAG_SKIP_PRECOMPUTE)
            print(f"{scheme:6s} {kind:8s} {ncol}x{nb}x{nz}{' f32' if f32 else ''}  kernel {ms:.4f} ms  frac {bps * ncol * nb / (ms * 1e-3) / 8e12:.3f}  K0 {k0[kind]:.4f} ms  "
                  f"{plans[kind].last_kernel()}", flush=True)
        del plans, out


if __name__ == "__main__":
    main()
