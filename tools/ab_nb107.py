"""Tridiagonal schemes at the reference's 107 bands: (M, T, store waves, kernel family) through crt_options.tune (tools; GPU box)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from crt1d_amd import _lib, batched, synth
d = synth.make_columns(30000, 107, 60)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
V = {"default": {}, "s1": {11: 1}, "s3": {11: 3}, "s4": {11: 4}, "M8": {8: 8}, "M16": {8: 16}, "M16 s3": {8: 16, 11: 3}, "M8 s3": {8: 8, 11: 3}, "T8 M16": {8: 16, 9: 8}, "k_tri_tile": {10: 1},
     "k_tri_tile M8 T8": {10: 1, 8: 8, 9: 8}, "generic flush": {13: 1}}
for scheme in ("n79", "zq", "zq_pa"):
    plan = batched.Plan(scheme, cols, bands)
    plan(); torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    res = {k: [] for k in V}; names = {}
    for rnd in range(3):
        for name, tune in V.items():
            plan.set_tune(tune)
            try:
                plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize(); names[name] = plan.last_kernel()
            except Exception as e:
                names[name] = "failed " + str(e)[:30]; res[name].append(float("nan")); continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
            e1.record(st); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 10)
    for name, v in res.items():
        v = sorted(v); print(f"{scheme} {name:18s} median {v[len(v)//2]:.4f} min {v[0]:.4f}  {names[name]}", flush=True)
    del plan
