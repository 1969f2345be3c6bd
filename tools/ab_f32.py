"""f32 storage: tile height / store waves of the closed-form pipeline, (M, T, store waves) of the tridiagonal ones (tools; GPU box)."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from crt1d_amd import _lib, batched, synth
F32 = ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r")
d = synth.make_columns(10000, 300, 60)
cols = batched.Columns.from_host(d)
b32 = batched.Bands.from_host({k: (d[k].astype(np.float32) if k in F32 else d[k]) for k in d})
CLOSED = {"default": {}, "T=4 s3": {4: 4, 3: 3}, "T=4 s2": {4: 4, 3: 2}, "T=4 s1": {4: 4, 3: 1}, "T=2 s2": {4: 2, 3: 2}, "T=8 s2": {4: 8, 3: 2}, "T=8 s4": {4: 8, 3: 4},
          "T=6 s3": {4: 6, 3: 3}, "T=16 s3": {4: 16, 3: 3}}
TRI = {"default": {}, "M12 T4 s2": {8: 12, 9: 4, 11: 2}, "M16 T4 s3": {8: 16, 9: 4, 11: 3}, "M16 T4 s2": {8: 16, 9: 4, 11: 2}, "M8 T4 s3": {8: 8, 9: 4, 11: 3}, "M8 T4 s2": {8: 8, 9: 4, 11: 2},
       "double-buffer M12": {10: 2, 8: 12}, "double-buffer M16 T8": {10: 2, 8: 16, 9: 8}, "M16 T8 s3": {8: 16, 9: 8, 11: 3}}
for scheme, variants in (("2s", CLOSED), ("bl", CLOSED), ("n79", TRI), ("zq", TRI)):
    plan = batched.Plan(scheme, cols, b32)
    plan(); torch.cuda.synchronize()
    res = {k: [] for k in variants}; names = {}
    st = torch.cuda.current_stream()
    for rnd in range(3):
        for name, tune in variants.items():
            plan.set_tune(tune)
            try:
                plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize(); names[name] = plan.last_kernel()
            except Exception as e:
                names[name] = "failed: " + str(e)[:40]; res[name].append(float("nan")); continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
            e1.record(st); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 10)
    for name, v in res.items():
        v = sorted(v); print(f"{scheme} f32 {name:22s} median {v[len(v)//2]:.4f} min {v[0]:.4f}  {names[name]}", flush=True)
    del plan
