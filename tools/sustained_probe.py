"""Is the 0.88 -> 0.83 step between 6-9 GB and 18-94 GB output sets a property of the SET or of how long the GPU has been busy?

    python tools/sustained_probe.py [scheme]

Per-launch HIP-event times of the solve kernel alone (K0 skipped), launches back to back:
  A  1e4 columns (5.9 GB set),  1500 launches  (~1.3 s of continuous work)
  B  125 000 columns (73.5 GB), 40 launches    (~0.45 s)
  C  the store-only probe of the same pattern on set B, 40 launches (no arithmetic: low power)
  D  A again after 2 s of idle
Printed: mean of launches [0,8), [8,16), ... so that a drift with busy time shows, whatever the set size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes

import torch

import bench
from crt1d_amd import _lib, batched, synth


def series(fn, n, st):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record(st)
    for i in range(n):
        fn()
        ev[i + 1].record(st)
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]


def show(tag, t, gbytes, group):
    cells = []
    for i in range(0, len(t), group):
        s = t[i:i + group]
        cells.append(f"{gbytes / (sum(s) / len(s) * 1e-3) / 8e3:.3f}")
    print(f"{tag}: frac of 8 TB/s per group of {group} launches: " + " ".join(cells), flush=True)


def main():
    scheme = sys.argv[1] if len(sys.argv) > 1 else "2s"
    nb, nz = 300, 60
    st = torch.cuda.current_stream()
    lib = _lib.load()
    bps = bench.bytes_per_solve(scheme, nz, 8)

    def plan_for(ncol):
        d = synth.make_columns(ncol, nb, nz, seed=1234)
        p = batched.Plan(scheme, batched.Columns.from_host(d), batched.Bands.from_host(d))
        p()
        torch.cuda.synchronize()
        return p

    pa = plan_for(10000)
    ga = bps * 10000 * nb / 1e9
    time.sleep(2.0)
    t = series(lambda: pa(st, flags=_lib.FLAG_SKIP_PRECOMPUTE), 1500, st)
    show("A  1e4 cols, cold start ", t[:64], ga, 8)
    show("A  ... whole run        ", t, ga, 100)
    pb = plan_for(125000)
    gb = bps * 125000 * nb / 1e9
    time.sleep(2.0)
    t = series(lambda: pb(st, flags=_lib.FLAG_SKIP_PRECOMPUTE), 40, st)
    show("B  125k cols, cold start", t, gb, 4)
    full = [pb.out[k] for k in ("I_dr", "I_df_d", "I_df_u", "F")]
    ptrs = (ctypes.c_void_p * len(full))(*[x.data_ptr() for x in full])
    time.sleep(2.0)
    t = series(lambda: lib.crt_hip_probe_store_set_f64(ptrs, len(full), 125000, nz * nb, 4 * nb, 0.5, st.cuda_stream), 40, st)
    show("C  store-only on set B  ", t, gb, 4)
    time.sleep(2.0)
    t = series(lambda: pa(st, flags=_lib.FLAG_SKIP_PRECOMPUTE), 600, st)
    show("D  1e4 cols after idle  ", t[:64], ga, 8)
    show("D  ... whole run        ", t, ga, 100)
    # E: the small plan right after the big one has kept the GPU busy for 0.45 s
    series(lambda: pb(st, flags=_lib.FLAG_SKIP_PRECOMPUTE), 40, st)
    t = series(lambda: pa(st, flags=_lib.FLAG_SKIP_PRECOMPUTE), 64, st)
    show("E  1e4 cols right after B", t, ga, 8)


if __name__ == "__main__":
    main()
