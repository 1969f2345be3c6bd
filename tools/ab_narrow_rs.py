"""Narrow band shards, register-staged pipeline: four staged pairs per store thread (eight workgroups per CU) vs two (ten / twelve), and
the checkpoint spacing (tools; GPU box)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from crt1d_amd import _lib, batched, synth
V = {"default": {}, "4 pairs": {2: 16}, "M=8": {8: 8}, "M=8 4 pairs": {8: 8, 2: 16}, "M=12": {8: 12}, "M=12 4 pairs": {8: 12, 2: 16}, "M=16": {8: 16}}
for shape in ((100000, 38, 100), (150000, 38, 60), (100000, 36, 100), (100000, 62, 60), (200000, 16, 60)):
    d = synth.make_columns(*shape)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    for scheme in ("zq", "n79"):
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
                for _ in range(5):
                    plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
                e1.record(st); torch.cuda.synchronize()
                res[name].append(e0.elapsed_time(e1) / 5)
        print(shape, scheme, {k: round(sorted(v)[1], 3) for k, v in res.items()}, "| default =", names["default"][11:75], flush=True)
        del plan
