"""Closed-form schemes at 107 / 128 / 200 bands: tile height and store waves of k_pipe with streaming stores (tools; GPU box)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from crt1d_amd import _lib, batched, synth
V = {"default": {}, "T=4 s1": {4: 4, 3: 1}, "T=4 s2": {4: 4, 3: 2}, "T=4 s3": {4: 4, 3: 3}, "T=8 s2": {4: 8, 3: 2}, "T=2 s2": {4: 2, 3: 2}, "T=6 s2": {4: 6, 3: 2}, "k_tile": {2: 4}}
for shape in ((30000, 107, 60), (25000, 128, 60), (15000, 200, 60), (12000, 256, 60)):
    d = synth.make_columns(*shape)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    for scheme in ("2s", "4s", "g77"):
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
                    names[name] = "failed"; res[name].append(float("nan")); continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(10):
                    plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
                e1.record(st); torch.cuda.synchronize()
                res[name].append(e0.elapsed_time(e1) / 10)
        print(shape, scheme, {k: round(sorted(v)[1], 3) for k, v in res.items()}, "|", names["default"][:62], flush=True)
        del plan
