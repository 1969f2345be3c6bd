"""f32 storage, closed-form pipeline: tile of 8 levels (two workgroups per CU) vs 4 levels (three) over schemes and shapes (tools; GPU box)."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from crt1d_amd import _lib, batched, synth
F32 = ("I_dr0", "I_df0", "leaf_r", "leaf_t", "soil_r")
for shape in ((10000, 300, 60), (3334, 300, 60), (6000, 300, 100), (20000, 200, 60)):
    d = synth.make_columns(*shape)
    cols = batched.Columns.from_host(d)
    b32 = batched.Bands.from_host({k: (d[k].astype(np.float32) if k in F32 else d[k]) for k in d})
    for scheme in ("2s", "4s", "bl"):
        plan = batched.Plan(scheme, cols, b32)
        plan(); torch.cuda.synchronize()
        st = torch.cuda.current_stream()
        out = []
        for name, tune in (("T=8", {4: 8}), ("T=4", {4: 4}), ("T=8", {4: 8}), ("T=4", {4: 4})):
            plan.set_tune(tune)
            plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
            e1.record(st); torch.cuda.synchronize()
            out.append((name, round(e0.elapsed_time(e1) / 10, 4)))
        print(shape, scheme, out, plan.last_kernel(), flush=True)
        del plan
