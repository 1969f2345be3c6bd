"""A/B of tile-kernel variants in ONE process, interleaved rounds (guide rule 24). Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import _lib, batched, synth

scheme = sys.argv[1] if len(sys.argv) > 1 else "2s"
ncol, nb, nz = 10000, 300, 60
d = synth.make_columns(ncol, nb, nz)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
plan = batched.Plan(scheme, cols, bands)
lib = _lib.load()
variants = {
    "tile T=8 fused": {0: 160 * 1024, 1: 8, 2: 0},
    "tile T=8 generic flush": {0: 160 * 1024, 1: 8, 2: 2},
    "tile T=4 fused": {0: 40960, 1: 4, 2: 0},
    "tile T=4 generic flush": {0: 40960, 1: 4, 2: 2},
    "tile T=12 fused": {0: 160 * 1024, 1: 12, 2: 0},
    "tile T=16 fused": {0: 160 * 1024, 1: 16, 2: 0},
    "direct": None,
}
res = {k: [] for k in variants}
st = torch.cuda.current_stream()
for rnd in range(6):
    for name, tune in variants.items():
        flags = _lib.FLAG_SKIP_PRECOMPUTE
        if tune is None:
            flags |= _lib.FLAG_DIRECT_STORES
        else:
            for k, v in tune.items():
                lib.crt_hip_tune(k, v)
        plan(flags=flags); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            plan(st, flags=flags)
        e1.record(st); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 10)
for name, v in res.items():
    v = sorted(v)
    print(f"{scheme} {name:24s} median {v[len(v)//2]:.4f} ms  min {v[0]:.4f} ms")
