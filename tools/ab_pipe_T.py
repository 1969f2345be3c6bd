import sys, os
sys.path.insert(0, "/root/repo")
import torch
from crt1d_amd import _lib, batched, synth
for scheme in ("4s", "2s", "bl"):
    d = synth.make_columns(10000, 300, 60)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    plan = batched.Plan(scheme, cols, bands)
    plan(); torch.cuda.synchronize()
    variants = {"default": {}, "T=2 s3": {4: 2, 3: 3}, "T=2 s2": {4: 2, 3: 2}, "T=4 s2": {4: 4, 3: 2}, "T=4 s4": {4: 4, 3: 4}, "T=4 s5": {4: 4, 3: 5}, "T=6 s3": {4: 6, 3: 3},
                "k_tile T=4": {2: 4}, "k_tile T=8": {2: 4, 1: 8}}
    res = {k: [] for k in variants}; names = {}
    st = torch.cuda.current_stream()
    for rnd in range(5):
        for name, tune in variants.items():
            plan.set_tune(tune)
            plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize(); names[name] = plan.last_kernel()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
            e1.record(st); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 10)
    for name, v in res.items():
        v = sorted(v); print(f"{scheme} {name:10s} median {v[len(v)//2]:.4f} min {v[0]:.4f}  {names[name]}")
    del plan
