"""A/B k_tile vs k_pipe for band counts away from the tuned 300 (odd nb, several columns per k_tile workgroup)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import _lib, batched, synth

lib = _lib.load()
st = torch.cuda.current_stream()
for scheme, ncol, nb, nz in [("2s", 30000, 107, 60), ("4s", 30000, 107, 60), ("g77", 30000, 107, 60), ("2s", 12000, 255, 60), ("2s", 10000, 301, 60), ("bl", 30000, 107, 60)]:
    d = synth.make_columns(ncol, nb, nz)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    plan = batched.Plan(scheme, cols, bands)
    plan(); torch.cuda.synchronize()
    variants = {"tile": {2: 4, 3: 0, 4: 0}, "pipe (auto)": {2: 0, 3: 0, 4: 0}, "pipe s3": {2: 0, 3: 3, 4: 0}, "pipe T4": {2: 0, 3: 0, 4: 4}, "pipe T16": {2: 0, 3: 0, 4: 16}}
    res = {k: [] for k in variants}
    for rnd in range(3):
        for name, tune in variants.items():
            plan.set_tune(tune)
            flags = _lib.FLAG_SKIP_PRECOMPUTE
            plan(flags=flags); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(5):
                plan(st, flags=flags)
            e1.record(st); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 5)
    gb = sum(v.numel() * 8 for v in plan.out.values()) / 1e9
    print(f"{scheme} {ncol}x{nb}x{nz} ({gb:.2f} GB): " + "  ".join(f"{k} {sorted(v)[len(v)//2]:.3f} ms = {gb / sorted(v)[len(v)//2]:.2f} TB/s" for k, v in res.items()), flush=True)
    del plan, cols, bands
    torch.cuda.empty_cache()
