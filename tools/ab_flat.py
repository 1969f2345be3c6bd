"""Odd band counts, tridiagonal schemes: per-array generic flush (tune 13 = 1) vs the fused flat flush.  One process, interleaved.
usage: python tools/ab_flat.py scheme ncol nb nz"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crt1d_amd import _lib, batched, synth

scheme = sys.argv[1]
ncol, nb, nz = (int(x) for x in sys.argv[2:5])
d = synth.make_columns(ncol, nb, nz, seed=1234)
cols, bands = batched.Columns.from_host(d), batched.Bands.from_host(d)
variants = {"generic (per array)": {13: 1}, "flat fused": {}, "flat, k_tri_tile": {10: 1}, "generic, k_tri_tile": {10: 1, 13: 1}, "flat, pipeline forced": {10: 4},
            "flat, pipe 1 store wave": {10: 4, 11: 1}, "flat, pipe 3 store waves": {10: 4, 11: 3}, "flat, pipe 4 store waves": {10: 4, 11: 4}}
plan = batched.Plan(scheme, cols, bands, placement="auto")
plan(flags=_lib.FLAG_DIRECT_STORES); torch.cuda.synchronize()
ref = {k: v.clone() for k, v in plan.out.items()}
st = torch.cuda.current_stream()
res, names = {k: [] for k in variants}, {}
for rnd in range(4):
    for name, tune in variants.items():
        plan.set_tune(tune)
        try:
            plan(flags=_lib.FLAG_SKIP_PRECOMPUTE); torch.cuda.synchronize()
        except RuntimeError as e:
            names[name] = f"unsupported ({e})"
            continue
        names[name] = plan.last_kernel()
        if rnd == 0:
            for k in ref:
                assert torch.equal(plan.out[k], ref[k]), (name, k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            plan(st, flags=_lib.FLAG_SKIP_PRECOMPUTE)
        e1.record(st); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 5)
gb = sum(v.numel() * 8 for v in plan.out.values()) / 1e9
print(f"{scheme} {ncol}x{nb}x{nz} ({gb:.2f} GB written)")
for name, v in res.items():
    if v:
        m = sorted(v)[len(v) // 2]
        print(f"  {name:26s} {m:8.3f} ms = {gb / m:5.2f} TB/s   {names[name]}")
    else:
        print(f"  {name:26s} {names.get(name)}")
