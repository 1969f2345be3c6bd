"""A/B of the tridiagonal column-tile kernel (M = checkpoint spacing, T = flush tile) vs the per-wave kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import _lib, batched, synth

scheme = sys.argv[1] if len(sys.argv) > 1 else "n79"
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ncol, nb = 10000, 300
d = synth.make_columns(ncol, nb, nz)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
plan = batched.Plan(scheme, cols, bands)
lib = _lib.load()
variants = {"auto": (0, 0), "M8 T4": (8, 4), "M12 T4": (12, 4), "M16 T4": (16, 4), "M8 T8": (8, 8), "M16 T8": (16, 8), "M12 T12": (12, 12), "M16 T8b": (16, 8),
            "M16 T16": (16, 16), "per-wave": None}
res = {k: [] for k in variants}
st = torch.cuda.current_stream()
for rnd in range(5):
    for name, mt in variants.items():
        flags = _lib.FLAG_SKIP_PRECOMPUTE
        if mt is None:
            flags |= _lib.FLAG_DIRECT_STORES
        else:
            lib.crt_hip_tune(8, mt[0]); lib.crt_hip_tune(9, mt[1])
        plan(flags=flags); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            plan(st, flags=flags)
        e1.record(st); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 5)
for name, v in res.items():
    v = sorted(v)
    print(f"{scheme} nz={nz} {name:10s} median {v[len(v)//2]:.4f} ms  min {v[0]:.4f} ms")
