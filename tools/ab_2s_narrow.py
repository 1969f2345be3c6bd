import sys
sys.path.insert(0, "/root/repo")
import torch
from crt1d_amd import _lib, batched, synth
V = {"default": {}, "T=4": {4: 4}, "T=4 s1": {4: 4, 3: 1}, "T=8 s1": {4: 8, 3: 1}, "T=4 s3": {4: 4, 3: 3}, "T=16 s2": {4: 16, 3: 2}}
for shape in ((200000, 38, 60), (100000, 38, 100), (400000, 16, 60)):
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
                plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize(); names[name] = plan.last_kernel()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(5):
                    plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
                e1.record(st); torch.cuda.synchronize()
                res[name].append(e0.elapsed_time(e1) / 5)
        print(shape, scheme, {k: round(sorted(v)[1], 3) for k, v in res.items()}, "|", names["default"][:60], flush=True)
        del plan
