"""A/B for the tridiagonal schemes at odd band counts: k_tri_tile (generic flush) vs k_tri_pipe (generic flush by store waves)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import _lib, batched, synth

lib = _lib.load()
st = torch.cuda.current_stream()
for scheme, ncol, nb, nz in [("n79", 30000, 107, 60), ("zq", 30000, 107, 60), ("n79", 12000, 255, 60), ("zq", 12000, 255, 60), ("zq", 18000, 107, 100)]:
    d = synth.make_columns(ncol, nb, nz)
    cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
    plan = batched.Plan(scheme, cols, bands)
    plan(); torch.cuda.synchronize()
    variants = {"tile": (1, 0), "pipe s1": (4, 1), "pipe s2": (4, 2), "pipe s3": (4, 3)}
    res = {k: [] for k in variants}
    for rnd in range(4):
        for name, (k10, k11) in variants.items():
            plan.set_tune({10: k10, 11: k11})
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
