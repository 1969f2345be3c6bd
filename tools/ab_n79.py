"""n79 at 1e4 x 300 x 60 (and nz = 100): pipeline variants through crt_options.tune (tools; GPU box)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from crt1d_amd import _lib, batched, synth
V = {"default": {}, "s2": {11: 2}, "s4": {11: 4}, "M16": {8: 16}, "M16 s4": {8: 16, 11: 4}, "M8": {8: 8}, "double-buffer": {10: 2}, "double-buffer M16": {10: 2, 8: 16}, "M16 T8": {8: 16, 9: 8}}
for shape in ((10000, 300, 60), (6000, 300, 100)):
    d = synth.make_columns(*shape)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    for scheme in ("n79", "zq"):
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
            v = sorted(v); print(f"{shape} {scheme} {name:18s} median {v[len(v)//2]:.4f} min {v[0]:.4f}  {names[name][11:]}", flush=True)
        del plan
